// testbed.cpp — see testbed.h.  Host orchestration only: every device-side computation goes through include/ngp_hip.h.
#include "testbed.h"
#include "snapshot.h"
#include "nerf_loader.h"
#include "dp.h"
#include "nerf_renderer.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <map>

// Events of the training step order work between streams of ONE device (march on stream B -> network pass on stream A, grid gradients
// -> the all-reduce stream) or are waited on by a host that reads host-coherent pinned words a kernel fenced itself (post_words).
// A device-scope release is enough for both and saves the system-scope cache write-back the default record does (step -1 %).
static constexpr unsigned STEP_EVENT_FLAGS = hipEventDisableTiming | hipEventReleaseToDevice;

namespace ngp {

static constexpr uint32_t GRID_CELLS = NGP_NERF_GRID_N_CELLS;
static constexpr float LOSS_SCALE = 128.0f;                    // testbed.h:272
static constexpr float NERF_MIN_OPTICAL_THICKNESS = 0.01f;     // testbed_nerf.cu:68
static constexpr uint32_t BATCH_SIZE_GRANULARITY = 128;        // tcnn::batch_size_granularity
static constexpr uint32_t OUT_STRIDE = 4;                      // fp16 (r,g,b,sigma) per sample; the reference pads to 16
static constexpr uint32_t MARCH_ITER = 10000;                  // testbed_nerf.cu:70

static inline uint32_t next_multiple(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }
static inline void hip_check(hipError_t e, const char* what) {
	if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_CHECK_THROW(x) hip_check((x), #x)

// ------------------------------------------------------------------------------------------------ DeviceBuffer
static std::atomic<size_t> g_total_allocated{0};
DeviceBuffer::~DeviceBuffer() { try { free(); } catch (...) {} }
DeviceBuffer& DeviceBuffer::operator=(DeviceBuffer&& o) noexcept {
	if (this != &o) { try { free(); } catch (...) {} m_ptr = o.m_ptr; m_bytes = o.m_bytes; o.m_ptr = nullptr; o.m_bytes = 0; }
	return *this;
}
void DeviceBuffer::free() {
	if (m_ptr) { (void)hipFree(m_ptr); g_total_allocated -= m_bytes; m_ptr = nullptr; m_bytes = 0; }
}
void DeviceBuffer::resize(size_t bytes) {
	if (bytes == m_bytes) return;
	free();
	if (bytes) { HIP_CHECK_THROW(hipMalloc(&m_ptr, bytes)); m_bytes = bytes; g_total_allocated += bytes; }
}
void DeviceBuffer::enlarge(size_t bytes) { if (bytes > m_bytes) resize(bytes); }
void DeviceBuffer::memset(int value, void* stream) { if (m_bytes) HIP_CHECK_THROW(hipMemsetAsync(m_ptr, value, m_bytes, (hipStream_t)stream)); }
void DeviceBuffer::copy_from_host(const void* src, size_t bytes, size_t dst_offset) {
	if (dst_offset + bytes > m_bytes) throw std::runtime_error("DeviceBuffer::copy_from_host out of range");
	if (bytes) HIP_CHECK_THROW(hipMemcpy((char*)m_ptr + dst_offset, src, bytes, hipMemcpyHostToDevice));
}
void DeviceBuffer::copy_to_host(void* dst, size_t bytes, size_t src_offset) const {
	if (src_offset + bytes > m_bytes) throw std::runtime_error("DeviceBuffer::copy_to_host out of range");
	if (bytes) HIP_CHECK_THROW(hipMemcpy(dst, (const char*)m_ptr + src_offset, bytes, hipMemcpyDeviceToHost));
}
size_t DeviceBuffer::total_allocated() { return g_total_allocated; }

// ------------------------------------------------------------------------------------------------ dataset
Mat34 NerfDataset::nerf_matrix_to_ngp(const Mat34& in, bool scale_columns) const {
	Mat34 r = in;
	const float s0 = scale_columns ? scale : 1.f, s12 = scale_columns ? -scale : -1.f;
	for (int k = 0; k < 3; ++k) { r.m[k] *= s0; r.m[3 + k] *= s12; r.m[6 + k] *= s12; }
	r.m[9] = r.m[9] * scale + offset.x; r.m[10] = r.m[10] * scale + offset.y; r.m[11] = r.m[11] * scale + offset.z;
	if (from_mitsuba) {
		for (int k = 0; k < 3; ++k) { r.m[k] *= -1.f; r.m[6 + k] *= -1.f; }
	} else {
		// cycle axes xyz <- yzx (rows)
		for (int c = 0; c < 4; ++c) { float t = r.m[3 * c]; r.m[3 * c] = r.m[3 * c + 1]; r.m[3 * c + 1] = r.m[3 * c + 2]; r.m[3 * c + 2] = t; }
	}
	return r;
}
Mat34 NerfDataset::ngp_matrix_to_nerf(const Mat34& in, bool scale_columns) const {
	Mat34 r = in;
	if (from_mitsuba) {
		for (int k = 0; k < 3; ++k) { r.m[k] *= -1.f; r.m[6 + k] *= -1.f; }
	} else {
		for (int c = 0; c < 4; ++c) { float t = r.m[3 * c + 2]; r.m[3 * c + 2] = r.m[3 * c + 1]; r.m[3 * c + 1] = r.m[3 * c]; r.m[3 * c] = t; }
	}
	const float s0 = scale_columns ? 1.f / scale : 1.f, s12 = scale_columns ? -1.f / scale : -1.f;
	for (int k = 0; k < 3; ++k) { r.m[k] *= s0; r.m[3 + k] *= s12; r.m[6 + k] *= s12; }
	r.m[9] = (r.m[9] - offset.x) / scale; r.m[10] = (r.m[10] - offset.y) / scale; r.m[11] = (r.m[11] - offset.z) / scale;
	return r;
}
void NerfDataset::set_training_image(int frame_idx, int w, int h, const void* pixels_host, int image_data_type, const float* depth_host, float depth_scale) {
	if (frame_idx < 0 || (size_t)frame_idx >= n_images) throw std::runtime_error{"NerfDataset::set_training_image: invalid frame index"};
	const size_t px = (size_t)w * h;
	const size_t stride = image_data_type == 1 ? 4 : (image_data_type == 2 ? 8 : 16);
	pixelmemory[frame_idx].resize(px * stride);
	pixelmemory[frame_idx].copy_from_host(pixels_host, px * stride);
	sharpness_valid = false;
	NgpImageMeta& m = metadata[frame_idx];
	m.pixels = pixelmemory[frame_idx].data();
	m.image_data_type = image_data_type;
	m.res[0] = w; m.res[1] = h;
	// depth (nerf_loader.cu:785-802, copy_depth 91-100): stored pre-multiplied by depth_scale; scale < 0 frees it, no data or scale 0 zeroes it
	if (depthmemory.size() < n_images) depthmemory.resize(n_images);
	if (depth_scale >= 0.f) {
		std::vector<float> scaled(px, 0.f);
		if (depth_host && depth_scale > 0.f) for (size_t k = 0; k < px; ++k) scaled[k] = depth_host[k] * depth_scale;
		depthmemory[frame_idx].resize(px * 4);
		depthmemory[frame_idx].copy_from_host(scaled.data(), px * 4);
		m.depth = depthmemory[frame_idx].as<float>();
	} else {
		depthmemory[frame_idx] = DeviceBuffer{};
		m.depth = nullptr;
	}
	update_metadata(frame_idx, frame_idx + 1);
}
void NerfDataset::update_sharpness() {   // nerf_loader.cu:829-834, for every image
	if (sharpness_valid || n_images == 0) return;
	const size_t per = (size_t)sharpness_resolution[0] * sharpness_resolution[1];
	sharpness_data.resize(per * n_images * 4);
	sharpness_data.memset(0, nullptr);
	for (size_t i = 0; i < n_images; ++i) {
		const NgpImageMeta& m = metadata[i];
		if (!m.pixels) continue;
		if (ngp_hip_compute_sharpness(nullptr, sharpness_resolution, m.res, m.pixels, m.image_data_type, sharpness_data.as<float>() + per * i) != 0)
			throw std::runtime_error{std::string{"compute_sharpness failed: "} + ngp_hip_last_error()};
	}
	if (hipDeviceSynchronize() != hipSuccess) throw std::runtime_error{"compute_sharpness: device error"};
	sharpness_valid = true;
}

void NerfDataset::sharpen_training_image(int frame_idx, float sharpen_amount) {
	sharpness_valid = false;
	if (!(sharpen_amount > 0.f)) return;
	if (frame_idx < 0 || (size_t)frame_idx >= n_images) throw std::runtime_error{"NerfDataset::sharpen_training_image: invalid frame index"};
	NgpImageMeta& m = metadata[frame_idx];
	const uint64_t px = (uint64_t)m.res[0] * m.res[1];
	auto chk = [](int rc, const char* what) { if (rc) throw std::runtime_error{std::string{what} + ": " + ngp_hip_last_error()}; };
	if (m.image_data_type == 1) {   // Byte -> premultiplied linear half4 (the hot-pink mask colour of the converted bytes becomes -1)
		DeviceBuffer half4; half4.resize(px * 8);
		chk(ngp_hip_image_from_rgba32_f16(nullptr, px, pixelmemory[frame_idx].as<uint8_t>(), half4.as<uint16_t>(), 0x00FF00FFu), "image_from_rgba32_f16");
		HIP_CHECK_THROW(hipStreamSynchronize(nullptr));
		pixelmemory[frame_idx] = std::move(half4);
		m.image_data_type = 2;
	}
	DeviceBuffer sharpened; sharpened.resize(px * (m.image_data_type == 2 ? 8 : 16));
	chk(ngp_hip_image_sharpen(nullptr, px, (uint32_t)m.res[0], pixelmemory[frame_idx].data(), sharpened.data(), m.image_data_type, sharpen_amount), "image_sharpen");
	HIP_CHECK_THROW(hipStreamSynchronize(nullptr));
	pixelmemory[frame_idx] = std::move(sharpened);
	m.pixels = pixelmemory[frame_idx].data();
	update_metadata(frame_idx, frame_idx + 1);
}
void NerfDataset::update_metadata(int first, int last) {
	if (last < 0 || last > (int)n_images) last = (int)n_images;
	int n = last - first;
	if (n <= 0) return;
	metadata_gpu.enlarge(n_images * sizeof(NgpImageMeta));
	metadata_gpu.copy_from_host(metadata.data() + first, (size_t)n * sizeof(NgpImageMeta), (size_t)first * sizeof(NgpImageMeta));
}

// ------------------------------------------------------------------------------------------------ training-side setters
void NerfTraining::set_image(int frame_idx, int w, int h, const float* rgba_host, const float* depth_host, float depth_scale) {
	if (owner) owner->invalidate_training_inputs();
	if (frame_idx < 0 || (size_t)frame_idx >= dataset.n_images) throw std::runtime_error{"Invalid frame index"};
	dataset.set_training_image(frame_idx, w, h, rgba_host, 3, depth_host, depth_scale);
}
void NerfTraining::set_image_rgba8(int frame_idx, int w, int h, const uint8_t* rgba_host) {
	if (owner) owner->invalidate_training_inputs();
	if (frame_idx < 0 || (size_t)frame_idx >= dataset.n_images) throw std::runtime_error{"Invalid frame index"};
	dataset.set_training_image(frame_idx, w, h, rgba_host, 1);
}
void NerfTraining::set_camera_extrinsics(int frame_idx, const Mat34& camera_to_world, bool convert_to_ngp) {
	if (frame_idx < 0 || (size_t)frame_idx >= dataset.n_images) return;
	if (owner) owner->invalidate_training_inputs();
	Mat34 m = convert_to_ngp ? dataset.nerf_matrix_to_ngp(camera_to_world) : camera_to_world;
	memcpy(dataset.xforms[frame_idx].start, m.m, sizeof(m.m));
	memcpy(dataset.xforms[frame_idx].end, m.m, sizeof(m.m));
	memset(dataset.metadata[frame_idx].rolling_shutter, 0, sizeof(float) * 4);
	dataset.update_metadata(frame_idx, frame_idx + 1);
	if ((size_t)frame_idx < cam_rot_offset.size()) cam_rot_offset[frame_idx].reset_state();   // 2533-2535
	if ((size_t)frame_idx < cam_pos_offset.size()) cam_pos_offset[frame_idx].reset_state();
	if ((size_t)frame_idx < cam_exposure.size()) cam_exposure[frame_idx] = ExposureAdam{};
	update_transforms(frame_idx, frame_idx + 1);
}
Mat34 NerfTraining::get_camera_extrinsics(int frame_idx) const {
	if (frame_idx < 0 || (size_t)frame_idx >= dataset.n_images) return Mat34{};
	Mat34 m; memcpy(m.m, transforms[frame_idx].start, sizeof(m.m));
	return dataset.ngp_matrix_to_nerf(m);
}
void NerfTraining::set_camera_intrinsics(int frame_idx, float fx, float fy, float cx, float cy, float k1, float k2, float p1, float p2) {
	if (frame_idx < 0 || (size_t)frame_idx >= dataset.n_images) return;
	if (owner) owner->invalidate_training_inputs();
	if (fx <= 0.f) fx = fy;
	if (fy <= 0.f) fy = fx;
	NgpImageMeta& m = dataset.metadata[frame_idx];
	if (cx < 0.f) cx = -cx; else cx = cx / (float)m.res[0];
	if (cy < 0.f) cy = -cy; else cy = cy / (float)m.res[1];
	m.lens_mode = (k1 || k2 || p1 || p2) ? 1 : 0;
	memset(m.lens_params, 0, sizeof(m.lens_params));
	m.lens_params[0] = k1; m.lens_params[1] = k2; m.lens_params[2] = p1; m.lens_params[3] = p2;
	m.principal_point[0] = cx; m.principal_point[1] = cy;
	m.focal_length[0] = fx; m.focal_length[1] = fy;
	dataset.update_metadata(frame_idx, frame_idx + 1);
}
void NerfTraining::reset_camera_extrinsics() {
	for (auto& o : cam_rot_offset) o.reset_state();
	for (auto& o : cam_pos_offset) o.reset_state();
	for (auto& o : cam_exposure) o = ExposureAdam{};
}
void NerfTraining::update_transforms(int first, int last) {
	if (last < 0 || last > (int)dataset.n_images) last = (int)dataset.n_images;
	int n = last - first;
	if (n <= 0) return;
	if (transforms.size() < (size_t)last) transforms.resize(last);
	for (int i = first; i < last; ++i) {
		NgpXForm xform = dataset.xforms[i];
		if ((size_t)i < cam_rot_offset.size()) {   // 2614-2622: the angle-axis offset rotates both shutter matrices
			const float* rot = cam_rot_offset[i].variable;
			const float angle = std::sqrt(rot[0] * rot[0] + rot[1] * rot[1] + rot[2] * rot[2]);
			if (angle > 0) {
				const float axis[3] = {rot[0] / angle, rot[1] / angle, rot[2] / angle};
				float r[9];
				angle_axis_to_matrix(angle, axis, r);
				mat3_mul(r, xform.start, xform.start);   // the rotation block is the first 9 floats of the column-major 3x4
				mat3_mul(r, xform.end, xform.end);
			}
		}
		if ((size_t)i < cam_pos_offset.size()) {   // 2624-2627
			for (int c = 0; c < 3; ++c) { xform.start[9 + c] += cam_pos_offset[i].variable[c]; xform.end[9 + c] += cam_pos_offset[i].variable[c]; }
		}
		transforms[i] = xform;
	}
	transforms_gpu.enlarge(dataset.n_images * sizeof(NgpXForm));
	transforms_gpu.copy_from_host(transforms.data() + first, (size_t)n * sizeof(NgpXForm), (size_t)first * sizeof(NgpXForm));
}

void RenderBuffer::resize(int w, int h) {
	if (w == res[0] && h == res[1]) return;
	res[0] = w; res[1] = h;
	const size_t n = (size_t)w * h;
	frame_buffer.resize(n * 16); depth_buffer.resize(n * 4); accumulate_buffer.resize(n * 16); surface.resize(n * 16);
	reset_accumulation();
}

// ------------------------------------------------------------------------------------------------ Testbed
// ---- one stream pair per device, shared by every Testbed of the process ---------------------------------------------------
// HIP spreads a process's streams over FOUR hardware queues in creation order.  With a stream pair per Testbed, a second live instance's training stream and run-ahead
// stream could land on ONE queue — its march then runs inside the chain instead of beside it (fox: 0.72 -> 0.79 ms per step next to an idle lego Testbed, found in round 4;
// GPU_MAX_HW_QUEUES=8 cured it, an environment variable the user had to know).  A Testbed's work is stream-ordered and two Testbeds of one process share the chip
// anyway, so all of them queue on the same two streams: the chain stream and the run-ahead stream are two streams created back to back — two queues — whoever uses them.
namespace {
struct StreamPair { hipStream_t a = nullptr, b = nullptr; int refs = 0; };
std::mutex g_stream_pair_mutex;
std::map<int, StreamPair> g_stream_pairs;
void acquire_stream_pair(int device, void*& a, void*& b) {
	std::lock_guard<std::mutex> lock(g_stream_pair_mutex);
	StreamPair& p = g_stream_pairs[device];
	if (p.refs == 0) {
		HIP_CHECK_THROW(hipStreamCreate(&p.a));
		if (hipStreamCreate(&p.b) != hipSuccess) { (void)hipStreamDestroy(p.a); p.a = nullptr; throw std::runtime_error{"hipStreamCreate failed"}; }
	}
	++p.refs;
	a = p.a; b = p.b;
}
void release_stream_pair(int device) {
	std::lock_guard<std::mutex> lock(g_stream_pair_mutex);
	auto it = g_stream_pairs.find(device);
	if (it == g_stream_pairs.end() || it->second.refs == 0) return;
	if (--it->second.refs == 0) {
		(void)hipStreamDestroy(it->second.b); (void)hipStreamDestroy(it->second.a);
		g_stream_pairs.erase(it);
	}
}
}  // namespace

Testbed::Testbed(ETestbedMode mode) : m_testbed_mode(mode) {
	if (mode == ETestbedMode::Volume) {
		throw std::runtime_error{"TestbedMode.Volume is outside the scope of this build (SURVEY.md §8): Nerf, and the plumbing configs Image / Sdf"};
	}
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
		throw std::runtime_error{"no MI355X / ROCm device visible: the product path has no CPU fallback"};
	}
	HIP_CHECK_THROW(hipGetDevice(&m_device));   // the caller's current device (one process per GPU: torch.cuda.set_device(LOCAL_RANK) came first); worker threads re-select it
	acquire_stream_pair(m_device, m_stream, m_stream_b);
	m_nerf.training.owner = this;
	m_rng = Pcg32(m_seed);
	reset_camera();
	m_network_config = Json::object();
}

Testbed::~Testbed() {
	try { shutdown_data_parallel(); } catch (...) {}
	bl_wait_for_renders();   // before any stream goes away (a Python owner releases the GIL around this wait: python_api.cpp TestbedDeleter)
	for (void* st : m_render_streams) { (void)hipStreamSynchronize((hipStream_t)st); (void)hipStreamDestroy((hipStream_t)st); }
	if (m_render_host_words) (void)hipHostFree(m_render_host_words);
	if (m_render_event) (void)hipEventDestroy((hipEvent_t)m_render_event);
	(void)hipDeviceSynchronize();
	if (m_pinned) (void)hipHostFree(m_pinned);
	if (m_host_words) (void)hipHostFree(m_host_words);
	if (m_counters_event) (void)hipEventDestroy((hipEvent_t)m_counters_event);
	if (m_prefetch_event) (void)hipEventDestroy((hipEvent_t)m_prefetch_event);
	if (m_grid_prefetch_event) (void)hipEventDestroy((hipEvent_t)m_grid_prefetch_event);
	if (m_adam_event) (void)hipEventDestroy((hipEvent_t)m_adam_event);
	if (m_ema_event) (void)hipEventDestroy((hipEvent_t)m_ema_event);
	if (m_stream_b) (void)hipStreamSynchronize((hipStream_t)m_stream_b);
	if (m_stream) (void)hipStreamSynchronize((hipStream_t)m_stream);
	if (m_stream) release_stream_pair(m_device);
}

void Testbed::check(int rc, const char* what) {
	if (rc != 0) throw std::runtime_error(std::string(what) + " failed: " + ngp_hip_last_error());
	if (m_trace_sync) {   // debugging aid (pyngp: trace_sync): drain both streams behind every launch group and say which one it was (finds a kernel that never returns)
		fprintf(stderr, "[ngp] step %u: %s ...", m_training_step, what); fflush(stderr);
		if (m_stream_b) (void)hipStreamSynchronize((hipStream_t)m_stream_b);
		if (m_stream) (void)hipStreamSynchronize((hipStream_t)m_stream);
		fprintf(stderr, " done\n"); fflush(stderr);
	}
}
float Testbed::backward_live_fraction() {
	if (!m_live_last_batch || m_live_count.bytes() == 0) return 1.0f;
	sync();
	uint32_t words[2] = {0u, 0u};
	m_live_count.copy_to_host(words, 8);
	return (float)words[m_live_last_parity] / (float)m_live_last_batch;
}
// (readers on another thread — the worker of request_nerf_render_async — come through here too: the flag is atomic and the wait + clear is one critical section with
// the record + set of optimizer_step, so a stage is never left un-joined and the event is never waited on while it is being re-recorded)
void Testbed::join_side_ema() {
	if (!m_ema_pending.load(std::memory_order_acquire)) return;
	std::lock_guard<std::mutex> lock(m_ema_mutex);
	if (!m_ema_pending.load(std::memory_order_relaxed)) return;
	HIP_CHECK_THROW(hipStreamWaitEvent((hipStream_t)m_stream, (hipEvent_t)m_ema_event, 0));
	m_ema_pending.store(false, std::memory_order_release);
}
void Testbed::sync() { join_side_ema(); HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream)); }
void Testbed::invalidate_training_inputs() { drop_prefetch(); ++m_state_version; }

// ---- live kernel timing ----------------------------------------------------------------------------------------
void* Testbed::prof_event() {
	if (!m_prof_event_pool.empty()) { void* e = m_prof_event_pool.back(); m_prof_event_pool.pop_back(); return e; }
	hipEvent_t e;
	HIP_CHECK_THROW(hipEventCreate(&e));
	return e;
}
void Testbed::profile_begin(int k, void* stream) {
	if (!m_profile_enabled || !((m_profile_mask >> k) & 1u)) return;
	if (m_profile_every > 1 && m_training_step % m_profile_every) return;   // profile_end finds no pending bracket for this launch and returns
	ProfPending p{k, prof_event(), nullptr, 0};
	HIP_CHECK_THROW(hipEventRecord((hipEvent_t)p.e0, (hipStream_t)(stream ? stream : m_stream)));
	m_prof_pending.push_back(p);
}
void Testbed::profile_end(int k, uint64_t units, void* stream) {
	if (!m_profile_enabled || !((m_profile_mask >> k) & 1u)) return;
	for (auto it = m_prof_pending.rbegin(); it != m_prof_pending.rend(); ++it) {
		if (it->k == k && !it->e1) {
			it->e1 = prof_event();
			it->units = units;
			HIP_CHECK_THROW(hipEventRecord((hipEvent_t)it->e1, (hipStream_t)(stream ? stream : m_stream)));
			return;
		}
	}
}
void Testbed::profile_collect(bool only_finished) {
	std::vector<ProfPending> keep;
	for (auto& p : m_prof_pending) {
		if (only_finished && (!p.e1 || hipEventQuery((hipEvent_t)p.e1) != hipSuccess)) { keep.push_back(p); continue; }
		if (p.e0 && p.e1) {
			(void)hipEventSynchronize((hipEvent_t)p.e1);
			float ms = 0.f;
			if (hipEventElapsedTime(&ms, (hipEvent_t)p.e0, (hipEvent_t)p.e1) == hipSuccess) {
				m_prof[p.k].ms += ms; m_prof[p.k].launches += 1; m_prof[p.k].units += p.units;
			}
		}
		if (p.e0) m_prof_event_pool.push_back(p.e0);
		if (p.e1) m_prof_event_pool.push_back(p.e1);
	}
	m_prof_pending.swap(keep);
}
void Testbed::reset_profile() {
	profile_collect();
	for (auto& a : m_prof) a = ProfAccum{};
}

// ---- network-pass organisation: measured per workload (testbed.h ENetworkPass) ------------------------------------
void Testbed::tuner_reset() {
	NetworkPassTuner& t = m_pass_tuner;
	for (auto& p : t.pending) { if (p.e0) m_prof_event_pool.push_back(p.e0); if (p.e1) m_prof_event_pool.push_back(p.e1); }
	t = NetworkPassTuner{};
	// NeRF: the first 16 steps update the whole occupancy grid every step and the ray batch is still growing towards the target; Image / Sdf have no such phase
	t.next_calibration_step = m_testbed_mode == ETestbedMode::Nerf ? 48u : 8u;
}
void Testbed::tuner_collect(bool wait) {
	NetworkPassTuner& t = m_pass_tuner;
	size_t kept = 0;
	for (auto& p : t.pending) {
		if (!wait && (!p.e1 || hipEventQuery((hipEvent_t)p.e1) != hipSuccess)) { t.pending[kept++] = p; continue; }
		float ms = 0.f;
		if (p.e1 && hipEventSynchronize((hipEvent_t)p.e1) == hipSuccess && hipEventElapsedTime(&ms, (hipEvent_t)p.e0, (hipEvent_t)p.e1) == hipSuccess && t.count[p.org] < NetworkPassTuner::N_SAMPLES)
			t.us[p.org][t.count[p.org]++] = ms * 1000.f;
		if (p.e0) m_prof_event_pool.push_back(p.e0);
		if (p.e1) m_prof_event_pool.push_back(p.e1);
	}
	t.pending.resize(kept);
	if (t.remaining == 0 && t.pending.empty() && t.count[0] && t.count[1]) {   // a calibration is complete: decide
		float med[2];
		for (int o = 0; o < 2; ++o) { std::sort(t.us[o], t.us[o] + t.count[o]); med[o] = t.us[o][t.count[o] / 2]; }
		const ENetworkPass other = t.chosen == ENetworkPass::Fused ? ENetworkPass::TwoKernel : ENetworkPass::Fused;
		const float cur = med[t.chosen == ENetworkPass::TwoKernel], alt = med[other == ENetworkPass::TwoKernel];
		if (alt < 0.97f * cur) t.chosen = other;
		t.last_us[0] = med[0]; t.last_us[1] = med[1];
		t.count[0] = t.count[1] = 0;
		++t.n_calibrations;
	}
}
bool Testbed::tuner_pick(void* stream) {
	NetworkPassTuner& t = m_pass_tuner;
	t.measuring = -1;
	// a bracket whose closing event was never recorded — the launch between tuner_pick and tuner_done threw — would sit in `pending` for ever and with it the
	// calibration (tuner_collect keeps entries without e1): drop such entries here, their sample is simply not taken
	for (size_t i = 0; i < t.pending.size();) {
		if (t.pending[i].e1) { ++i; continue; }
		if (t.pending[i].e0) m_prof_event_pool.push_back(t.pending[i].e0);
		t.pending.erase(t.pending.begin() + (long)i);
	}
	if (m_network_pass != ENetworkPass::Auto) return m_network_pass == ENetworkPass::TwoKernel;
	if (!t.pending.empty() || t.count[0] || t.count[1]) tuner_collect(false);
	if (t.remaining == 0 && t.pending.empty() && m_training_step >= t.next_calibration_step) {
		t.remaining = 2 * NetworkPassTuner::N_SAMPLES;
		t.count[0] = t.count[1] = 0;
		t.last_calibration_step = m_training_step;
		// 48 -> 320 (the update cadence has reached every 16th step, the batch its size) -> 1056 -> every 4096 steps
		t.next_calibration_step = m_training_step < 300u ? 320u : m_training_step < 1000u ? 1056u : m_training_step + 4096u;
	}
	if (t.remaining > 0) {
		t.measuring = t.remaining & 1;
		--t.remaining;
		NetworkPassTuner::Pending p{prof_event(), nullptr, t.measuring};
		HIP_CHECK_THROW(hipEventRecord((hipEvent_t)p.e0, (hipStream_t)stream));
		t.pending.push_back(p);
		return t.measuring == 1;
	}
	return t.chosen == ENetworkPass::TwoKernel;
}
void Testbed::tuner_done(void* stream) {
	NetworkPassTuner& t = m_pass_tuner;
	if (t.measuring < 0 || t.pending.empty()) return;
	NetworkPassTuner::Pending& p = t.pending.back();
	p.e1 = prof_event();
	HIP_CHECK_THROW(hipEventRecord((hipEvent_t)p.e1, (hipStream_t)stream));
	t.measuring = -1;
}

void Testbed::reset_camera() {  // testbed.cu:283-299
	m_fov_axis = 1;
	set_fov(50.625f);
	m_zoom = 1.f;
	m_screen_center[0] = m_screen_center[1] = 0.5f;
	m_scale = 1.5f;
	const float c[12] = {1, 0, 0, 0, -1, 0, 0, 0, -1, 0.5f, 0.5f, 0.5f};
	memcpy(m_camera.m, c, sizeof(c));
	// m_camera.col(3) -= m_scale * view_dir(), view_dir = col(2)
	for (int k = 0; k < 3; ++k) m_camera.m[9 + k] -= m_scale * m_camera.m[6 + k];
}
float Testbed::fov() const { return 2.f * 180.f / 3.14159265358979323846f * atanf(1.0f / (m_relative_focal_length[m_fov_axis] * 2.f)); }
void Testbed::set_fov(float val) {
	const float f = 0.5f * 1.0f / tanf(0.5f * val * 3.14159265358979323846f / 180.f);
	m_relative_focal_length[0] = m_relative_focal_length[1] = f;
}
void Testbed::fov_xy(float out[2]) const { for (int k = 0; k < 2; ++k) out[k] = 2.f * 180.f / 3.14159265358979323846f * atanf(1.0f / (m_relative_focal_length[k] * 2.f)); }
void Testbed::set_fov_xy(const float val[2]) { for (int k = 0; k < 2; ++k) m_relative_focal_length[k] = 0.5f * 1.0f / tanf(0.5f * val[k] * 3.14159265358979323846f / 180.f); }

static inline Vec3 v3_add(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline Vec3 v3_sub(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 v3_mul(Vec3 a, float s) { return Vec3{a.x * s, a.y * s, a.z * s}; }
static inline float v3_dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline Vec3 v3_cross(Vec3 a, Vec3 b) { return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline Vec3 v3_normalized(Vec3 a) { const float n = std::sqrt(v3_dot(a, a)); return n > 0.f ? v3_mul(a, 1.f / n) : a; }

// position lerp + rotation slerp between two column-major 3x4 poses (Eigen: Quaternionf(R0).slerp(t, Quaternionf(R1)).normalized().toRotationMatrix())
static void mat_to_quat(const float* m, float q[4]) {   // q = (w, x, y, z); R(r, c) = m[3 c + r]
	auto R = [&](int r, int c) { return m[3 * c + r]; };
	const float tr = R(0, 0) + R(1, 1) + R(2, 2);
	if (tr > 0.f) {
		float t = std::sqrt(tr + 1.0f); q[0] = 0.5f * t; t = 0.5f / t;
		q[1] = (R(2, 1) - R(1, 2)) * t; q[2] = (R(0, 2) - R(2, 0)) * t; q[3] = (R(1, 0) - R(0, 1)) * t;
	} else {
		int i = 0; if (R(1, 1) > R(0, 0)) i = 1; if (R(2, 2) > R(i, i)) i = 2;
		const int j = (i + 1) % 3, k = (j + 1) % 3;
		float t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0f);
		q[1 + i] = 0.5f * t; t = 0.5f / t;
		q[0] = (R(k, j) - R(j, k)) * t; q[1 + j] = (R(j, i) + R(i, j)) * t; q[1 + k] = (R(k, i) + R(i, k)) * t;
	}
}
static Mat34 pose_lerp(const float* p0, const float* p1, float t) {
	Mat34 rv;
	if (t == 0.f || memcmp(p0, p1, sizeof(float) * 12) == 0) { memcpy(rv.m, p0, sizeof(rv.m)); return rv; }
	float a[4], b[4], q[4];
	mat_to_quat(p0, a); mat_to_quat(p1, b);
	const float d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3], ad = std::fabs(d);
	float s0, s1;
	if (ad >= 1.0f - 1.1920929e-07f) { s0 = 1.0f - t; s1 = t; }
	else { const float th = std::acos(ad), st = std::sin(th); s0 = std::sin((1.0f - t) * th) / st; s1 = std::sin(t * th) / st; }
	if (d < 0.f) s1 = -s1;
	for (int k = 0; k < 4; ++k) q[k] = s0 * a[k] + s1 * b[k];
	const float n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
	for (int k = 0; k < 4; ++k) q[k] /= n;
	const float w = q[0], x = q[1], y = q[2], z = q[3];
	const float Rm[9] = {1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w),      // column 0
	                     2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w),      // column 1
	                     2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)};     // column 2
	memcpy(rv.m, Rm, sizeof(Rm));
	for (int k = 0; k < 3; ++k) rv.m[9 + k] = p0[9 + k] + (p1[9 + k] - p0[9 + k]) * t;
	return rv;
}

Vec3 Testbed::look_at() const { return v3_add(view_pos(), v3_mul(view_dir(), m_scale)); }                       // testbed.cu:223-225
void Testbed::set_look_at(const Vec3& pos) { const Vec3 d = v3_sub(pos, look_at()); m_camera.m[9] += d.x; m_camera.m[10] += d.y; m_camera.m[11] += d.z; }
void Testbed::set_scale(float scale) {                                                                          // testbed.cu:231-235
	const Vec3 prev = look_at();
	const Vec3 p = v3_add(v3_mul(v3_sub(view_pos(), prev), scale / m_scale), prev);
	m_camera.m[9] = p.x; m_camera.m[10] = p.y; m_camera.m[11] = p.z;
	m_scale = scale;
}
void Testbed::set_view_dir(const Vec3& dir) {                                                                    // testbed.cu:237-243
	const Vec3 old = look_at();
	const Vec3 c0 = v3_normalized(v3_cross(dir, m_up_dir)), c1 = v3_normalized(v3_cross(dir, c0)), c2 = v3_normalized(dir);
	const Vec3 cols[3] = {c0, c1, c2};
	for (int c = 0; c < 3; ++c) { m_camera.m[3 * c] = cols[c].x; m_camera.m[3 * c + 1] = cols[c].y; m_camera.m[3 * c + 2] = cols[c].z; }
	set_look_at(old);
}
void Testbed::set_camera_to_training_view(int trainview) {                                                       // testbed.cu:273-281
	NerfTraining& tr = m_nerf.training;
	if (trainview < 0 || (size_t)trainview >= tr.dataset.n_images) throw std::runtime_error{"Invalid training view"};
	const Vec3 old_look_at = look_at();
	const NgpImageMeta& m = tr.dataset.metadata[trainview];
	// get_xform_given_rolling_shutter(xform, rolling_shutter, uv = (0.5, 0.5), motionblur_time = 0) (common_device.cuh:224-234)
	const float t = m.rolling_shutter[0] + m.rolling_shutter[1] * 0.5f + m.rolling_shutter[2] * 0.5f;
	m_camera = pose_lerp(tr.transforms[trainview].start, tr.transforms[trainview].end, t);
	m_relative_focal_length[0] = m.focal_length[0] / (float)m.res[m_fov_axis]; m_relative_focal_length[1] = m.focal_length[1] / (float)m.res[m_fov_axis];
	m_scale = std::max(v3_dot(v3_sub(old_look_at, view_pos()), view_dir()), 0.1f);
	m_nerf.render_with_lens_distortion = true;
	m_nerf.render_lens_proxy = m;
	const NgpImageMeta& m0 = tr.dataset.metadata[0];
	m_screen_center[0] = 1.f - m0.principal_point[0]; m_screen_center[1] = 1.f - m0.principal_point[1];
}
void Testbed::first_training_view() { m_nerf.training.view = 0; set_camera_to_training_view(0); m_windowless_render_surface.reset_accumulation(); }
void Testbed::last_training_view() { m_nerf.training.view = (int)m_nerf.training.dataset.n_images - 1; set_camera_to_training_view(m_nerf.training.view); m_windowless_render_surface.reset_accumulation(); }
void Testbed::previous_training_view() { if (m_nerf.training.view != 0) m_nerf.training.view -= 1; set_camera_to_training_view(m_nerf.training.view); m_windowless_render_surface.reset_accumulation(); }
void Testbed::next_training_view() { if (m_nerf.training.view != (int)m_nerf.training.dataset.n_images - 1) m_nerf.training.view += 1; set_camera_to_training_view(m_nerf.training.view); m_windowless_render_surface.reset_accumulation(); }

// m_render_aabb_to_local is a column-major 3x3 (element (r, c) at [3 c + r]), like the Eigen matrix it replaces
Mat34 Testbed::crop_box(bool nerf_space) const {                                                                  // testbed.cu:395-410
	const float* L = m_render_aabb_to_local;
	const float c[3] = {0.5f * (m_render_aabb.min[0] + m_render_aabb.max[0]), 0.5f * (m_render_aabb.min[1] + m_render_aabb.max[1]), 0.5f * (m_render_aabb.min[2] + m_render_aabb.max[2])};
	const float rad[3] = {0.5f * (m_render_aabb.max[0] - m_render_aabb.min[0]), 0.5f * (m_render_aabb.max[1] - m_render_aabb.min[1]), 0.5f * (m_render_aabb.max[2] - m_render_aabb.min[2])};
	Mat34 rv;
	for (int k = 0; k < 3; ++k) {
		rv.m[9 + k] = L[3 * k + 0] * c[0] + L[3 * k + 1] * c[1] + L[3 * k + 2] * c[2];   // cen = L^T * center
		for (int ax = 0; ax < 3; ++ax) rv.m[3 * ax + k] = L[3 * k + ax] * rad[ax];         // column ax = row ax of L, times the radius
	}
	if (nerf_space) rv = m_nerf.training.dataset.ngp_matrix_to_nerf(rv, true);
	return rv;
}
void Testbed::set_crop_box(Mat34 m, bool nerf_space) {                                                            // testbed.cu:412-424
	if (nerf_space) m = m_nerf.training.dataset.nerf_matrix_to_ngp(m, true);
	float rad[3];
	for (int ax = 0; ax < 3; ++ax) rad[ax] = std::sqrt(m.m[3 * ax] * m.m[3 * ax] + m.m[3 * ax + 1] * m.m[3 * ax + 1] + m.m[3 * ax + 2] * m.m[3 * ax + 2]);
	float* L = m_render_aabb_to_local;
	for (int ax = 0; ax < 3; ++ax) for (int k = 0; k < 3; ++k) L[3 * k + ax] = m.m[3 * ax + k] / rad[ax];   // row ax of L = column ax of m / radius
	for (int r = 0; r < 3; ++r) {
		const float cen = L[r] * m.m[9] + L[3 + r] * m.m[10] + L[6 + r] * m.m[11];
		m_render_aabb.min[r] = cen - rad[r]; m_render_aabb.max[r] = cen + rad[r];
	}
}
std::vector<Vec3> Testbed::crop_box_corners(bool nerf_space) const {                                             // testbed.cu:426-445
	const Mat34 m = crop_box(nerf_space);
	std::vector<Vec3> rv(8);
	for (int i = 0; i < 8; ++i) {
		const float v[3] = {(i & 1) ? 1.f : -1.f, (i & 2) ? 1.f : -1.f, (i & 4) ? 1.f : -1.f};
		rv[i] = Vec3{m.m[0] * v[0] + m.m[3] * v[1] + m.m[6] * v[2] + m.m[9], m.m[1] * v[0] + m.m[4] * v[1] + m.m[7] * v[2] + m.m[10], m.m[2] * v[0] + m.m[5] * v[1] + m.m[8] * v[2] + m.m[11]};
	}
	return rv;
}

void Testbed::load_training_data(const std::string& data_path) {  // testbed.cu:196-218 (Nerf mode) -> Testbed::load_nerf (testbed_nerf.cu:2735-2759)
	if (m_testbed_mode == ETestbedMode::Image) { load_image(data_path); return; }
	if (m_testbed_mode == ETestbedMode::Sdf) throw std::runtime_error{"mesh loading (OBJ/STL + BVH) is outside the scope of this build: feed (position, distance) pairs with override_sdf_training_data"};
	drop_prefetch();
	++m_state_version;
	if (data_path.size() > 8 && data_path.substr(data_path.size() - 8) == ".msgpack") { load_snapshot(data_path); m_train = false; return; }
	const LoadedNerfData data = load_nerf_host(resolve_nerf_json_paths(data_path), m_nerf.sharpen);
	NerfDataset& d = m_nerf.training.dataset;
	d = NerfDataset{};
	d.n_images = data.n_images;
	d.paths = data.paths; d.xforms = data.xforms; d.metadata = data.metadata;
	d.pixelmemory.clear(); d.pixelmemory.resize(d.n_images);
	d.scale = data.scale; d.offset = data.offset; d.aabb_scale = data.aabb_scale; d.from_mitsuba = data.from_mitsuba; d.is_hdr = data.is_hdr;
	d.render_aabb = data.render_aabb; d.up = data.up;
	d.n_extra_learnable_dims = data.n_extra_learnable_dims; d.has_light_dirs = data.has_light_dirs; d.light_dirs = data.light_dirs;
	d.envmap_data = data.envmap_data; d.envmap_resolution[0] = data.envmap_resolution[0]; d.envmap_resolution[1] = data.envmap_resolution[1];
	d.has_rays = data.has_rays;
	d.raymemory.clear(); d.raymemory.resize(d.n_images);
	for (size_t i = 0; i < d.n_images; ++i) {
		const size_t px = (size_t)data.metadata[i].res[0] * data.metadata[i].res[1];
		std::vector<float> depth;
		if (!data.depth16[i].empty()) { depth.resize(px); for (size_t k = 0; k < px; ++k) depth[k] = (float)data.depth16[i][k]; }   // copy_depth<uint16_t> (nerf_loader.cu:91-100)
		d.set_training_image((int)i, data.metadata[i].res[0], data.metadata[i].res[1], data.pixels[i].data(), data.image_type[i], depth.empty() ? nullptr : depth.data(),
		                     data.depth_scale[i] * data.scale /* 727 */);
		d.sharpen_training_image((int)i, data.sharpen_amount);
		if (!data.rays[i].empty()) {
			d.raymemory[i].resize(px * sizeof(NgpRay));
			d.raymemory[i].copy_from_host(data.rays[i].data(), px * sizeof(NgpRay));
			d.metadata[i].rays = d.raymemory[i].as<NgpRay>();
		}
	}
	d.update_metadata();
	m_data_path = data_path;
	load_nerf_post();
	m_training_data_available = true;
}

void Testbed::create_empty_nerf_dataset(size_t n_images, int aabb_scale, bool is_hdr) {  // testbed_nerf.cu:2635-2641, nerf_loader.cu:175-195
	NerfDataset& d = m_nerf.training.dataset;
	d = NerfDataset{};
	d.n_images = n_images;
	d.xforms.resize(n_images);
	d.metadata.assign(n_images, NgpImageMeta{});
	d.pixelmemory.clear(); d.pixelmemory.resize(n_images);
	d.aabb_scale = aabb_scale;
	d.is_hdr = is_hdr;
	for (size_t i = 0; i < n_images; ++i) {
		const float ident[12] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0};
		memcpy(d.xforms[i].start, ident, sizeof(ident));
		memcpy(d.xforms[i].end, ident, sizeof(ident));
		d.metadata[i].principal_point[0] = d.metadata[i].principal_point[1] = 0.5f;
		d.metadata[i].focal_length[0] = d.metadata[i].focal_length[1] = 1000.f;
	}
	load_nerf_post();
	m_nerf.training.n_images_for_training = 0;
	m_training_data_available = true;
}

void Testbed::load_nerf_post() {
	NerfTraining& tr = m_nerf.training;
	m_nerf.rgb_activation = tr.dataset.is_hdr ? ENerfActivation::Exponential : ENerfActivation::Logistic;
	tr.n_images_for_training = (int)tr.dataset.n_images;
	tr.dataset.update_metadata();
	std::vector<float> zeros(std::max<size_t>(tr.dataset.n_images, 1) * 3, 0.f);
	tr.cam_exposure_gpu.resize(zeros.size() * 4);
	tr.cam_exposure_gpu.copy_from_host(zeros.data(), zeros.size() * 4);
	if (tr.dataset.has_rays) tr.near_distance = 0.0f;
	tr.cam_pos_offset.assign(tr.dataset.n_images, Vec3Adam{});      // 2654-2655 (learning rates are set per update, 3076-3077)
	tr.cam_rot_offset.assign(tr.dataset.n_images, RotationAdam{});
	tr.update_transforms();
	if (!tr.dataset.metadata.empty()) {
		m_nerf.render_lens_proxy = tr.dataset.metadata[0];
		m_screen_center[0] = 1.f - tr.dataset.metadata[0].principal_point[0];
		m_screen_center[1] = 1.f - tr.dataset.metadata[0].principal_point[1];
	}
	const int s = tr.dataset.aabb_scale;
	if (s <= 0 || (s & (s - 1))) throw std::runtime_error{"NeRF dataset's `aabb_scale` must be a power of two, but is " + std::to_string(s) + "."};
	const int max_aabb_scale = 1 << (NGP_NERF_CASCADES - 1);
	if (s > max_aabb_scale) throw std::runtime_error{"NeRF dataset must have `aabb_scale <= " + std::to_string(max_aabb_scale) + "`, but is " + std::to_string(s) + "."};
	const float half = 0.5f * (float)std::min(max_aabb_scale, s);
	for (int k = 0; k < 3; ++k) { m_aabb.min[k] = 0.5f - half; m_aabb.max[k] = 0.5f + half; }
	m_raw_aabb = m_aabb;
	m_render_aabb = m_aabb;
	const NgpAabb& ra = tr.dataset.render_aabb;
	if (!(ra.max[0] < ra.min[0] || ra.max[1] < ra.min[1] || ra.max[2] < ra.min[2])) {
		for (int k = 0; k < 3; ++k) { m_render_aabb.min[k] = std::max(ra.min[k], m_aabb.min[k]); m_render_aabb.max[k] = std::min(ra.max[k], m_aabb.max[k]); }
	}
	m_nerf.max_cascade = 0;
	while ((1 << m_nerf.max_cascade) < s) ++m_nerf.max_cascade;
	m_nerf.cone_angle_constant = s <= 1 ? 0.0f : (1.0f / 256.0f);
}

// ---- network config ---------------------------------------------------------------------------------------------
Json Testbed::load_network_config(const std::string& path) {  // testbed.cu:120-145
	return load_config_or_snapshot(path);
}
void Testbed::reload_network_from_file(const std::string& path) {
	if (!path.empty()) m_network_config_path = path;
	m_network_config = load_network_config(m_network_config_path);
	reset_network();
}
void Testbed::reload_network_from_json(const Json& json, const std::string& config_base_path) {
	m_network_config_path = config_base_path;
	m_network_config = json;
	reset_network();
}

static ELossType string_to_loss_type(std::string s) {  // testbed.cu:2220-2240
	std::transform(s.begin(), s.end(), s.begin(), ::tolower);
	if (s == "l2") return ELossType::L2;
	if (s == "relativel2") return ELossType::RelativeL2;
	if (s == "l1") return ELossType::L1;
	if (s == "mape") return ELossType::Mape;
	if (s == "smape") return ELossType::Smape;
	if (s == "huber" || s == "smoothl1") return ELossType::Huber;
	if (s == "logl1") return ELossType::LogL1;
	throw std::runtime_error{"Unknown loss type."};
}

void Testbed::parse_optimizer_config(const Json& opt_in) {
	// configs/nerf/base.json:5-22: Ema{decay} o ExponentialDecay{decay_start, decay_interval, decay_base} o Adam{...}; any subset of the wrappers
	m_use_ema = false; m_has_decay = false;
	m_ema_decay = 0.95f; m_decay_start = 0; m_decay_interval = 1; m_decay_end = 0; m_decay_base = 1.0f;
	const Json* o = &opt_in;
	while (true) {
		std::string otype = o->value("otype", "Adam");
		std::string lower = otype; std::transform(lower.begin(), lower.end(), lower.begin(), ::tolower);
		if (lower == "ema") { m_use_ema = true; m_ema_decay = (float)o->value("decay", 0.99); }
		else if (lower == "exponentialdecay") {
			m_has_decay = true;
			m_decay_start = (uint32_t)o->value("decay_start", 0); m_decay_interval = (uint32_t)o->value("decay_interval", 1);
			m_decay_end = (uint32_t)o->value("decay_end", 0); m_decay_base = (float)o->value("decay_base", 1.0);
		} else if (lower == "adam") {
			m_base_learning_rate = (float)o->value("learning_rate", 1e-3);
			m_beta1 = (float)o->value("beta1", 0.9); m_beta2 = (float)o->value("beta2", 0.999);
			m_epsilon = (float)o->value("epsilon", 1e-8); m_l2_reg = (float)o->value("l2_reg", 1e-8);
			break;
		} else {
			throw std::runtime_error{"optimizer otype '" + otype + "' is not part of the NeRF hot path (Ema / ExponentialDecay / Adam are)"};
		}
		if (!o->contains("nested")) throw std::runtime_error{"optimizer '" + otype + "' needs a nested optimizer"};
		o = &o->at("nested");
	}
	m_learning_rate = m_base_learning_rate;
}

// ---- TrainableBuffer: trainable_buffer.cuh + Trainer<float, float, float> + create_optimizer<float>(json) ----
void TrainableBuffer::reset(int w, int h, uint32_t dims, const Json& optimizer_config, void* stream) {
	resolution[0] = w; resolution[1] = h; n_dims = dims; step = 0;
	use_ema = false; has_decay = false; ema_decay = 0.99f; decay_start = 0; decay_interval = 1; decay_end = 0; decay_base = 1.0f;
	base_learning_rate = 1e-3f; beta1 = 0.9f; beta2 = 0.999f; epsilon = 1e-8f;
	const Json* o = optimizer_config.is_object() ? &optimizer_config : nullptr;
	while (o) {   // [Ema o] [ExponentialDecay o] Adam, as Testbed::parse_optimizer_config reads the network's
		std::string lower = o->value("otype", "Adam"); std::transform(lower.begin(), lower.end(), lower.begin(), ::tolower);
		if (lower == "ema") { use_ema = true; ema_decay = (float)o->value("decay", 0.99); }
		else if (lower == "exponentialdecay") {
			has_decay = true; decay_start = (uint32_t)o->value("decay_start", 0); decay_interval = (uint32_t)o->value("decay_interval", 1);
			decay_end = (uint32_t)o->value("decay_end", 0); decay_base = (float)o->value("decay_base", 1.0);
		} else if (lower == "adam") {
			base_learning_rate = (float)o->value("learning_rate", 1e-3); beta1 = (float)o->value("beta1", 0.9); beta2 = (float)o->value("beta2", 0.999); epsilon = (float)o->value("epsilon", 1e-8);
			break;
		} else throw std::runtime_error{"optimizer otype '" + lower + "' of a trainable buffer: Ema / ExponentialDecay / Adam are built"};
		o = o->contains("nested") ? &o->at("nested") : nullptr;
	}
	learning_rate = base_learning_rate;
	const size_t bytes = n_params() * sizeof(float);
	DeviceBuffer* all[6] = {&params, &ema, &gradients, &gradient_weights, &first_moments, &second_moments};
	for (DeviceBuffer* b : all) { b->resize(bytes); if (bytes) b->memset(0, stream); }
}
void TrainableBuffer::set_params(const float* host, size_t n) {
	if (n != n_params()) throw std::runtime_error{"TrainableBuffer::set_params: size mismatch"};
	params.copy_from_host(host, n * sizeof(float));
}
void TrainableBuffer::clear_gradients(void* stream, bool weights_too) {
	if (!n_params()) return;
	gradients.memset(0, stream);
	if (weights_too) gradient_weights.memset(0, stream);
}
void TrainableBuffer::optimizer_step(void* stream, float loss_scale) {
	if (!n_params()) return;
	++step;
	const int rc = ngp_hip_optimizer_step_f32(stream, (uint32_t)n_params(), step, learning_rate, beta1, beta2, epsilon, loss_scale, use_ema ? ema_decay : 0.0f, gradients.as<float>(),
	                                           params.as<float>(), first_moments.as<float>(), second_moments.as<float>(), use_ema ? ema.as<float>() : nullptr);
	if (rc != 0) throw std::runtime_error{std::string("ngp_hip_optimizer_step_f32: ") + ngp_hip_last_error()};
	if (has_decay && step >= decay_start && (decay_end == 0 || step < decay_end) && decay_interval && step % decay_interval == 0) learning_rate *= decay_base;   // as Testbed::optimizer_step
}

void Testbed::reset_network(bool clear_density_grid) {  // testbed.cu:2249-2470
	join_side_ema();
	if (m_testbed_mode != ETestbedMode::Nerf) { reset_network_gridmlp(); return; }
	drop_prefetch();
	if (m_live_count.bytes()) { m_live_count.memset(0, m_stream); m_live_parity = 0; m_live_last_batch = 0; }   // (a step that threw between the compaction and the backward pass may have left its counter behind)
	++m_state_version;
	m_n_matrix_params = NGP_MLP_N_PARAMS;
	m_rng = Pcg32(m_seed);
	m_windowless_render_surface.reset_accumulation();
	NerfTraining& tr = m_nerf.training;
	tr.counters_rgb.rays_per_batch = 1 << 12;
	tr.counters_rgb.measured_batch_size_before_compaction = 0;
	tr.n_steps_since_cam_update = 0;
	tr.n_steps_since_error_map_update = 0;
	tr.n_rays_since_error_map_update = 0;
	tr.n_steps_between_error_map_updates = 128;
	tr.density_grid_rng = Pcg32(m_rng.next_uint());
	tr.reset_camera_extrinsics();   // testbed.cu:2267: the optimizer states only — the transforms keep the old offsets until the next update_transforms, like there

	Json config = m_network_config;
	const Json empty = Json::object();
	const Json& enc = config.contains("encoding") ? config["encoding"] : empty;
	tr.loss_type = string_to_loss_type(config.contains("loss") ? config["loss"].value("otype", "L2") : std::string("L2"));

	// the fused kernels implement the base.json family: HashGrid L=16 F=2, 64-wide FullyFusedMLP 1+2 hidden layers, SH degree 4
	auto require = [](bool ok, const std::string& what) { if (!ok) throw std::runtime_error{"network config not supported by the gfx950 fused kernels: " + what}; };
	std::string enc_type = enc.value("otype", "HashGrid"); std::transform(enc_type.begin(), enc_type.end(), enc_type.begin(), ::tolower);
	require(enc_type == "hashgrid", "encoding.otype must be HashGrid");
	const uint32_t n_features_per_level = (uint32_t)enc.value("n_features_per_level", 2);
	require(n_features_per_level == 2, "n_features_per_level must be 2");
	m_num_levels = (uint32_t)enc.value("n_levels", 16);
	require(m_num_levels == 16, "n_levels must be 16");
	const uint32_t log2_hashmap_size = (uint32_t)enc.value("log2_hashmap_size", 15);
	m_base_grid_resolution = (uint32_t)enc.value("base_resolution", 0);
	if (!m_base_grid_resolution) m_base_grid_resolution = 1u << (log2_hashmap_size / 3);
	const float desired_resolution = 2048.0f;
	m_per_level_scale = (float)enc.value("per_level_scale", 0.0);
	if (m_per_level_scale <= 0.0f && m_num_levels > 1) {
		m_per_level_scale = std::exp(std::log(desired_resolution * (float)tr.dataset.aabb_scale / (float)m_base_grid_resolution) / (float)(m_num_levels - 1));
	}
	if (config.contains("network")) {
		require(config["network"].value("n_neurons", 64) == 64 && config["network"].value("n_hidden_layers", 1) == 1, "network must be 64 neurons x 1 hidden layer");
	}
	m_n_rgb_hidden_layers = 2;
	if (config.contains("rgb_network")) {   // configs/nerf/base_{0,1,2,3}layer.json: 0 hidden layers = one linear map ([tcnn] CutlassMLP), else 64-wide hidden layers
		const int h = config["rgb_network"].value("n_hidden_layers", 2);
		require(h >= 0 && h <= 3, "rgb_network.n_hidden_layers must be 0..3");
		require(h == 0 || config["rgb_network"].value("n_neurons", 64) == 64, "rgb_network must be 64 neurons wide");
		m_n_rgb_hidden_layers = (uint32_t)h;
	}
	// per-image extra dims (testbed.cu:2353-2357): the dataset decides
	m_n_extra_dims = tr.dataset.n_extra_dims();
	require(m_n_extra_dims <= 16, "at most 16 extra dims (n_extra_learnable_dims, or 3 light-direction dims)");
	check(ngp_hip_net_make_desc_host(m_num_levels, log2_hashmap_size, m_base_grid_resolution, m_per_level_scale, &m_desc), "ngp_hip_net_make_desc_host");
	NgpNetVariant nv;
	m_n_matrix_params = ngp_hip_net_mlp_params_host(net_variant(nv));
	m_n_params = (size_t)m_n_matrix_params + 2u * (size_t)m_desc.n_grid_entries;
	tr.reset_extra_dims(m_rng);   // testbed.cu:2347
	m_desc_gpu.resize(sizeof(NgpNetDesc));
	m_desc_gpu.copy_from_host(&m_desc, sizeof(NgpNetDesc));

	if (config.contains("optimizer")) parse_optimizer_config(config["optimizer"]);
	m_optimizer_step = 0;
	tuner_reset();
	m_dp_inference_stale = false;
	m_dp_state_stale = false;   // the whole fp32 state is rebuilt below (every rank of a live communicator resets alike: same seed, same bits)

	// (+ DP_PARAM_SLACK elements behind the weights and the gradients: the sharded optimizer step all-gathers world equal shards in place, the last one padded)
	m_params.resize((m_n_params + DP_PARAM_SLACK) * 2); m_inference_params.resize((m_n_params + DP_PARAM_SLACK) * 2); m_grads.resize((m_n_params + DP_PARAM_SLACK) * 2);
	m_master.resize((m_n_params + DP_PARAM_SLACK) * 4); m_first_moments.resize((m_n_params + DP_PARAM_SLACK) * 4); m_second_moments.resize((m_n_params + DP_PARAM_SLACK) * 4); m_ema.resize((m_n_params + DP_PARAM_SLACK) * 4);
	m_first_moments.memset(0, m_stream); m_second_moments.memset(0, m_stream); m_ema.memset(0, m_stream); m_grads.memset(0, m_stream); m_params.memset(0, m_stream); m_master.memset(0, m_stream);
	check(ngp_hip_nerf_init_params(m_stream, &m_desc, m_seed, m_master.as<float>(), m_params.as<uint16_t>(), m_inference_params.as<uint16_t>(), net_variant(nv)), "ngp_hip_nerf_init_params");

	{   // distortion map model (testbed.cu:2386-2396) and envmap model (2447-2462): their own optimizers, else the network's
		int dres[2] = {32, 32};
		const Json* dopt = config.contains("optimizer") ? &config["optimizer"] : nullptr;
		if (config.contains("distortion_map")) {
			const Json& dm = config["distortion_map"];
			if (dm.contains("optimizer")) dopt = &dm["optimizer"];
			if (dm.contains("resolution")) { dres[0] = (int)dm["resolution"][(size_t)0].number(); dres[1] = (int)dm["resolution"][(size_t)1].number(); }
		}
		m_distortion.reset(dres[0], dres[1], 2, dopt ? *dopt : Json{}, m_stream);
		const Json* eopt = config.contains("optimizer") ? &config["optimizer"] : nullptr;
		std::string eloss = config.contains("loss") ? config["loss"].value("otype", "L2") : std::string("L2");
		if (config.contains("envmap")) {
			const Json& em = config["envmap"];
			if (em.contains("optimizer")) eopt = &em["optimizer"];
			if (em.contains("loss")) eloss = em["loss"].value("otype", "L2");
		}
		const NerfDataset& ds = m_nerf.training.dataset;
		m_envmap.reset(ds.envmap_resolution[0], ds.envmap_resolution[1], 4, eopt ? *eopt : Json{}, m_stream);
		m_envmap.loss_type = string_to_loss_type(eloss);
		if (!ds.envmap_data.empty()) m_envmap.set_params(ds.envmap_data.data(), ds.envmap_data.size());
	}
	m_loss_scalar_gpu.resize(4);

	m_training_step = 0;
	m_loss_scalar = 0.f;
	if (clear_density_grid) {
		m_nerf.density_grid.resize((size_t)GRID_CELLS * (m_nerf.max_cascade + 1) * 4);
		m_nerf.density_grid.memset(0, m_stream);
		m_nerf.density_grid_bitfield.resize((size_t)GRID_CELLS);  // grid_mip_offset(NERF_CASCADES)/8
		m_nerf.density_grid_bitfield.memset(0, m_stream);
		m_nerf.brick_summary_valid = false;
		m_nerf.density_grid_mean.resize(4);
		m_nerf.density_grid_mean.memset(0, m_stream);
	}
	sync();
}

// ---- training ---------------------------------------------------------------------------------------------------
bool Testbed::frame() {  // testbed.cu:2044-2090 without the GUI: train_and_render(skip_rendering = true)
	if (m_train) train(m_training_batch_size);
	return true;
}

void Testbed::train(uint32_t batch_size) {  // testbed.cu:2527-2587
	if (!m_training_data_available) { m_train = false; return; }
	if (m_testbed_mode != ETestbedMode::Nerf) {
		if (m_n_params == 0) throw std::runtime_error{"train(): no network — call reload_network_from_file/json first"};
		const bool get_loss = m_training_step % 16 == 0;
		auto t0 = std::chrono::steady_clock::now();
		if (m_testbed_mode == ETestbedMode::Image) train_image(batch_size, get_loss); else train_sdf(batch_size, get_loss);
		sync();
		m_stats.training_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
		return;
	}
	if (m_n_params == 0) throw std::runtime_error{"train(): no network — call reload_network_from_file/json first"};
	m_windowless_render_surface.reset_accumulation();
	const uint32_t n_prep_to_skip = std::min(std::max(m_training_step / 16u, 1u), 16u);
	if (m_training_step % n_prep_to_skip == 0) {
		auto start = std::chrono::steady_clock::now();
		training_prep_nerf(batch_size);
		if (!m_async_training_steps) sync();   // (testbed.cu:2553 drains the stream here; nothing on the host reads what the update wrote, and the drain leaves the GPU idle for ~45 us)
		m_stats.training_prep_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - start).count() / n_prep_to_skip;
	}
	const bool get_loss_scalar = m_training_step % 16 == 0;
	auto start = std::chrono::steady_clock::now();
	train_nerf(batch_size, get_loss_scalar);
	if (m_async_training_steps) {
		// the reference drains the stream after every step (testbed.cu:2570).  Nothing in the step needs that: the host already waited
		// for the step's counters, everything later is ordered by the stream, and the drain leaves the GPU idle for the ~30 us until
		// the next step's first launch.  training_ms becomes the host time from one step's return to the next.
		if (m_profile_enabled && m_prof_pending.size() >= 64) profile_collect(true);
		const auto now = std::chrono::steady_clock::now();
		m_stats.training_ms = std::chrono::duration<float, std::milli>(now - (m_last_step_return.time_since_epoch().count() ? m_last_step_return : start)).count();
		m_last_step_return = now;
		return;
	}
	sync();
	if (m_profile_enabled) profile_collect();
	m_stats.training_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - start).count();
}

void Testbed::training_prep_nerf(uint32_t) {  // testbed_nerf.cu:3388-3401
	if (m_nerf.training.n_images_for_training == 0) return;
	const float alpha = m_nerf.training.density_grid_decay;
	const uint32_t n_cascades = m_nerf.max_cascade + 1;
	profile_begin(PK_GRID_PREP);
	const uint32_t n_grid_samples = m_training_step < 256 ? GRID_CELLS * n_cascades : GRID_CELLS / 2 * n_cascades;
	if (m_training_step < 256) update_density_grid_nerf(alpha, GRID_CELLS * n_cascades, 0);
	else update_density_grid_nerf(alpha, GRID_CELLS / 4 * n_cascades, GRID_CELLS / 4 * n_cascades);
	profile_end(PK_GRID_PREP, n_grid_samples);
}

void Testbed::update_density_grid_nerf(float decay, uint32_t n_uniform, uint32_t n_nonuniform) {  // testbed_nerf.cu:2761-2842
	NerfTraining& tr = m_nerf.training;
	const PrefetchedGridSamples gp = m_grid_prefetch;   // (taken over before drop_prefetch, which would drain stream B on the host for it)
	m_grid_prefetch.valid = false;
	drop_prefetch();     // the bitfield is about to change: samples marched ahead against the old one are void
	const bool grid_samples_ready = gp.valid && gp.step == m_training_step && gp.version == m_state_version && gp.n_uniform == n_uniform && gp.n_nonuniform == n_nonuniform &&
	                                gp.ema_step == m_nerf.density_grid_ema_step && gp.rng_state == tr.density_grid_rng.state && gp.rng_inc == tr.density_grid_rng.inc &&
	                                gp.n_images == tr.n_images_for_training && m_training_step != 0 && tr.n_images_for_training == tr.n_images_for_training_prev;
	if (gp.valid) HIP_CHECK_THROW(hipStreamWaitEvent((hipStream_t)m_stream, (hipEvent_t)m_grid_prefetch_event, 0));   // used or not: stream B must be through with the buffers
	++m_state_version;
	const uint32_t n_elements = GRID_CELLS * (m_nerf.max_cascade + 1);
	if (m_nerf.density_grid.bytes() != (size_t)n_elements * 4) { m_nerf.density_grid.resize((size_t)n_elements * 4); m_nerf.density_grid.memset(0, m_stream); }
	const uint32_t n_samples = n_uniform + n_nonuniform;
	m_grid_positions.enlarge((size_t)n_samples * 12); m_grid_indices.enlarge((size_t)n_samples * 4);
	m_grid_tmp.enlarge((size_t)n_elements * 4); m_grid_mlp_out.enlarge((size_t)n_samples * 2);
	float* grid = m_nerf.density_grid.as<float>();

	if (m_training_step == 0 || tr.n_images_for_training != tr.n_images_for_training_prev) {
		tr.n_images_for_training_prev = tr.n_images_for_training;
		if (m_training_step == 0) m_nerf.density_grid_ema_step = 0;
		if (!tr.dataset.has_rays) {
			check(ngp_hip_mark_untrained_density_grid(m_stream, n_elements, grid, (uint32_t)tr.n_images_for_training, tr.dataset.metadata_gpu.as<NgpImageMeta>(),
			                                          tr.transforms_gpu.as<NgpXForm>(), m_training_step == 0), "mark_untrained_density_grid");
		} else {
			m_nerf.density_grid.memset(0, m_stream);
		}
	}
	if (grid_samples_ready) { tr.density_grid_rng.advance(); tr.density_grid_rng.advance(); ++m_grid_prefetch_hits; }   // generated ahead on stream B (maybe_prefetch_grid_samples)
	else launch_grid_samples(m_stream, n_uniform, n_nonuniform);
	// density pass on the TRAINING weights (use_inference_params = false, testbed_nerf.cu:2833)
	m_enc_ws.enlarge(ngp_hip_nerf_encode_workspace_bytes(n_samples));
	NgpNetVariant nv_density;
	check(ngp_hip_nerf_density_ws(m_stream, m_desc_gpu.as<NgpNetDesc>(), m_params.as<uint16_t>(), m_grid_positions.as<float>(), 3, n_samples, m_grid_mlp_out.as<uint16_t>(), m_enc_ws.data(), m_enc_ws.bytes(), net_variant(nv_density)), "nerf_density");
	check(ngp_hip_splat_grid_samples_max(m_stream, n_samples, m_grid_indices.as<uint32_t>(), m_grid_mlp_out.as<uint16_t>(), m_grid_tmp.as<float>(), (int)m_nerf.density_activation), "splat");
	// the update's tail in one call: ema + partial means, bitfield (which sums the partials itself), the pooled levels (round 6: three launches and a memset less on the chain)
	m_nerf.density_grid_mean.enlarge(4);
	m_nerf.density_grid_bitfield.enlarge((size_t)GRID_CELLS);
	m_grid_mean_ws.enlarge((size_t)ngp_hip_density_grid_tail_workspace_bytes());
	check(ngp_hip_density_grid_ema_mean_bitfield(m_stream, m_nerf.max_cascade + 1, decay, grid, m_grid_tmp.as<float>(), m_nerf.density_grid_mean.as<float>(), m_nerf.density_grid_bitfield.as<uint8_t>(),
	                                             m_grid_mean_ws.data()), "density_grid_ema_mean_bitfield");
	++m_nerf.density_grid_ema_step;
	update_density_grid_mean_and_bitfield(true);
}

void Testbed::update_density_grid_mean_and_bitfield(bool bitfield_is_current) {  // testbed_nerf.cu:2844-2859
	if (!bitfield_is_current) {
		m_nerf.density_grid_bitfield.enlarge((size_t)GRID_CELLS);
		m_nerf.density_grid_mean.enlarge(4);
		check(ngp_hip_density_grid_mean(m_stream, m_nerf.density_grid.as<float>(), GRID_CELLS, m_nerf.density_grid_mean.as<float>()), "density_grid_mean");
		check(ngp_hip_grid_to_bitfield_and_pool(m_stream, m_nerf.density_grid.as<float>(), m_nerf.max_cascade + 1, m_nerf.density_grid_mean.as<float>(),
		                                        m_nerf.density_grid_bitfield.as<uint8_t>()), "grid_to_bitfield_and_pool");
	}
	m_nerf.bitfield_brick_summary.enlarge(GRID_CELLS / 64 / 32 * 4);
	check(ngp_hip_bitfield_brick_summary(m_stream, m_nerf.density_grid_bitfield.as<uint8_t>(), m_nerf.bitfield_brick_summary.as<uint32_t>()), "bitfield_brick_summary");
	m_nerf.brick_summary_valid = true;
}

void NerfTraining::reset_extra_dims(Pcg32& rng) {  // testbed_nerf.cu:2297-2318
	const uint32_t ne = dataset.n_extra_dims();
	extra_dims_opt.clear();
	if (ne == 0) return;
	std::vector<float> cpu((size_t)ne * (dataset.n_images + 1), 0.f);   // n_images + 1: the extra slot of the inference latent code
	extra_dims_opt.resize(dataset.n_images);
	for (size_t i = 0; i < dataset.n_images; ++i) {
		NerfTraining::ExtraDimsAdam& a = extra_dims_opt[i];
		a.iter = 0; a.m.assign(ne, 0.f); a.v.assign(ne, 0.f); a.x.assign(ne, 0.f);
		Vec3 ld{0.f, 0.f, 0.f};
		if (dataset.has_light_dirs && i < dataset.light_dirs.size()) {   // warp_direction(light_dir.normalized()) = (d + 1) / 2
			const Vec3 d = dataset.light_dirs[i];
			const float n = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
			if (n > 0.f) ld = Vec3{(d.x / n + 1.f) * 0.5f, (d.y / n + 1.f) * 0.5f, (d.z / n + 1.f) * 0.5f}; else ld = Vec3{0.5f, 0.5f, 0.5f};
		}
		const float l3[3] = {ld.x, ld.y, ld.z};
		for (uint32_t j = 0; j < ne; ++j) {
			const float v = (dataset.has_light_dirs && j < 3) ? l3[j] : rng.next_float() * 2.f - 1.f;
			cpu[i * ne + j] = v; a.x[j] = v;
		}
	}
	extra_dims_gpu.resize(cpu.size() * 4);
	extra_dims_gpu.copy_from_host(cpu.data(), cpu.size() * 4);
}

const float* Testbed::get_inference_extra_dims() {  // testbed_nerf.cu:2320-2337
	if (m_n_extra_dims == 0) return nullptr;
	NerfTraining& tr = m_nerf.training;
	const uint32_t ne = tr.dataset.n_extra_dims();
	if (ne != m_n_extra_dims || tr.extra_dims_gpu.bytes() < (size_t)ne * (tr.dataset.n_images + 1) * 4) throw std::runtime_error{"the dataset's extra dims changed after the network was built: call reload_network_from_file / reset_network"};
	const uint32_t idx = std::min<uint32_t>(m_nerf.extra_dim_idx_for_inference, tr.dataset.n_images ? (uint32_t)tr.dataset.n_images - 1 : 0u);
	const float* src = tr.extra_dims_gpu.as<float>() + (size_t)idx * ne;
	if (!tr.dataset.has_light_dirs) return src;
	// the dataset has light directions: the scratch slot behind the images' rows gets the chosen row with the requested light direction in front
	float* dst = tr.extra_dims_gpu.as<float>() + tr.dataset.n_images * ne;
	HIP_CHECK_THROW(hipMemcpyAsync(dst, src, (size_t)ne * 4, hipMemcpyDeviceToDevice, (hipStream_t)m_stream));
	const Vec3 d = m_nerf.light_dir;
	const float n = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
	const float ld[3] = {(d.x / n + 1.f) * 0.5f, (d.y / n + 1.f) * 0.5f, (d.z / n + 1.f) * 0.5f};
	HIP_CHECK_THROW(hipMemcpyAsync(dst, ld, std::min<size_t>((size_t)ne * 4, 12), hipMemcpyHostToDevice, (hipStream_t)m_stream));
	HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream));   // `ld` is a stack array
	return dst;
}

void Testbed::set_distributed(uint32_t rank, uint32_t world_size) {
	if (world_size == 0 || rank >= world_size) throw std::runtime_error{"set_distributed: bad rank / world_size"};
	drop_prefetch();
	++m_state_version;
	m_rank = rank; m_world_size = world_size;
}

void Testbed::init_data_parallel(uint32_t rank, uint32_t world_size, const std::string& key, bool strong_scaling) {
	shutdown_data_parallel();
	set_distributed(rank, world_size);
	m_dp_strong_scaling = strong_scaling;
	m_dp_exchange_step = 0;
	// (a world of one rank sets up the same machinery — a 1-rank segment and communicator — so that the whole step path can be run on one GPU)
	if (!ngp_rccl_available()) throw std::runtime_error{"init_data_parallel: no RCCL library could be loaded (librccl.so.1)"};
	m_dp_shm.reset(new ShmCounterExchange(rank, world_size, key));
	uint8_t id[128];
	if (rank == 0) { check(ngp_rccl_get_unique_id(id), "ngp_rccl_get_unique_id"); m_dp_shm->publish_blob(id); }
	else m_dp_shm->fetch_blob(id);
	m_dp_comm = ngp_rccl_init((int)rank, (int)world_size, id);
	if (!m_dp_comm) { m_dp_shm.reset(); set_distributed(0, 1); throw std::runtime_error{std::string{"ngp_rccl_init failed: "} + ngp_hip_last_error()}; }
	m_dp_shm->barrier();
	// strong scaling keeps the GLOBAL step of the single-GPU run: each rank marches 1 / world of the rays the counter feedback has settled on (at step 0: of the
	// reference's initial 4096, testbed.cu:2284), so the first data-parallel step is the single-rank step on the same rays instead of a world-times larger one that
	// overflows the per-rank sample buffer (tests/test_dp_loopback_gpu.py); the feedback carries on from there
	if (strong_scaling && world_size > 1) {
		NerfCounters& c = m_nerf.training.counters_rgb;
		c.rays_per_batch = std::max(next_multiple(c.rays_per_batch / world_size, BATCH_SIZE_GRANULARITY), BATCH_SIZE_GRANULARITY);
	}
	// with more than one rank the run-ahead march waits for the gradients and runs beside the exchange (maybe_prefetch_next): that needs the event behind the backward pass
	m_dp_march_behind_exchange = world_size > 1;
	if (m_dp_march_behind_exchange) m_want_grid_grad_event = true;
}
// The sharded optimizer step leaves the fp32 state (master weights, Adam moments) of other ranks' shards stale.  Whoever needs the whole state — a snapshot with
// optimizer state, training on after shutdown_data_parallel — gathers it first.  Collective: every rank of the communicator calls it.
void Testbed::dp_gather_optimizer_state() {
	join_side_ema();
	// a stale state whose communicator is gone cannot be made whole any more: saying "done" here would let training / save_snapshot run on fp32 state that is old outside
	// this rank's shard.  What rebuilds the whole state clears the flag instead: reset_network, load_snapshot.
	if (m_dp_state_stale && !m_dp_comm)
		throw std::runtime_error{"dp_gather_optimizer_state: the fp32 optimizer state is stale outside this rank's shard and the data-parallel communicator is gone — gather on ALL ranks BEFORE shutdown_data_parallel(); now only reset_network() / load_snapshot() rebuild the state"};
	if (m_dp_comm && m_world_size >= 2 && m_dp_sharded_optimizer && m_n_params != 0) {
		const uint32_t shard = next_multiple(((uint32_t)m_n_params + m_world_size - 1) / m_world_size, 8u);
		check(ngp_rccl_allgather_f32(m_dp_comm, m_stream, m_master.as<float>(), shard), "ngp_rccl_allgather_f32 (master weights)");
		check(ngp_rccl_allgather_f32(m_dp_comm, m_stream, m_first_moments.as<float>(), shard), "ngp_rccl_allgather_f32 (first moments)");
		check(ngp_rccl_allgather_f32(m_dp_comm, m_stream, m_second_moments.as<float>(), shard), "ngp_rccl_allgather_f32 (second moments)");
		if (m_dp_sharded_ema) check(ngp_rccl_allgather_f32(m_dp_comm, m_stream, m_ema.as<float>(), shard), "ngp_rccl_allgather_f32 (Ema)");
		sync();
	}
	m_dp_state_stale = false;   // (a world of one, or the replicated step, never left anything stale)
	dp_gather_inference_params();
}
void Testbed::set_dp_sharded_optimizer(bool on) {
	if (on == m_dp_sharded_optimizer) return;
	// switching under a live communicator would either run the replicated step on fp32 state that is stale outside this rank's shard (sharded -> replicated) or
	// need a collective inside a property setter: choose before init_data_parallel, or gather (all ranks) and shut the communicator down first
	if (m_dp_comm) throw std::runtime_error{"dp_sharded_optimizer can only be changed before init_data_parallel (or after dp_gather_optimizer_state() + shutdown_data_parallel())"};
	m_dp_sharded_optimizer = on;
}
void Testbed::set_dp_sharded_ema(bool on) {
	if (on == m_dp_sharded_ema) return;
	if (m_dp_comm) throw std::runtime_error{"dp_sharded_ema can only be changed before init_data_parallel (or after dp_gather_optimizer_state() + shutdown_data_parallel())"};
	m_dp_sharded_ema = on;
}
// No collective here: a rank that leaves alone (an exception, a test that ends) must not hang in an all-gather the others never enter.  A Testbed that is to train
// on, or to save its optimizer state, after sharded data-parallel steps calls dp_gather_optimizer_state() on ALL ranks first; otherwise the state stays marked stale.
void Testbed::shutdown_data_parallel() {
	if (m_dp_comm) { if (m_stream) (void)hipStreamSynchronize((hipStream_t)m_stream); ngp_rccl_finalize(m_dp_comm); m_dp_comm = nullptr; }
	m_dp_shm.reset();
}

void Testbed::train_nerf(uint32_t target_batch_size, bool get_loss_scalar) {  // testbed_nerf.cu:2896-3023
	if (m_nerf.training.n_images_for_training == 0) return;
	// refuse before anything is queued (the check used to sit behind the step's forward and loss launches)
	if ((m_nerf.training.optimize_extrinsics || m_nerf.training.optimize_distortion) && !net_is_base_family() && m_netx_scalar_kernels)
		throw std::runtime_error{"optimize_extrinsics / optimize_distortion need the network's input gradient, which the scalar checker kernels (netx_scalar_kernels) do not have"};
	if (m_dp_comm) {
		// the data-parallel step: every rank marches its slice of the step's rays; {samples, compacted samples, loss} are summed over
		// the ranks right behind the loss kernel (hosts, shared memory), the gradient vector between backward and optimizer (RCCL, stream order);
		// optimizer and occupancy-grid updates run replicated on identical parameters and rng, i.e. bit-identically — no collective for them
		uint32_t B = target_batch_size;
		if (m_dp_strong_scaling) {
			if (target_batch_size % (256 * m_world_size)) throw std::runtime_error{"strong scaling: the batch size must be a multiple of 256 x world_size"};
			B = target_batch_size / m_world_size;
		}
		uint32_t counters[2];
		train_nerf_dp_begin(B, counters, get_loss_scalar);
		const double mine[3] = {(double)counters[0], (double)counters[1], get_loss_scalar ? (double)local_loss_sum() : 0.0};
		double sum[3];
		m_dp_shm->all_sum(m_dp_exchange_step++, mine, sum);
		train_nerf_dp_backward(B, (uint32_t)sum[0], (uint32_t)sum[1], get_loss_scalar, (float)sum[2]);
		if (!m_dp_sharded_optimizer) {   // the round-1 / round-2 exchange: fp16 all-reduce of the whole gradient vector, the optimizer step replicated on every rank
			profile_begin(PK_GRAD_EXCHANGE);
			check(ngp_rccl_allreduce_grads(m_dp_comm, m_stream, m_grads.as<uint16_t>(), m_n_params), "ngp_rccl_allreduce_grads");
			profile_end(PK_GRAD_EXCHANGE, m_n_params);
		}
		train_nerf_dp_end();
		return;
	}
	if (m_world_size != 1) throw std::runtime_error{"train(): world_size > 1 — drive the step with train_nerf_dp_begin / _dp_backward / _dp_end around the all-reduces"};
	uint32_t counters[2];
	train_nerf_dp_begin(target_batch_size, counters, get_loss_scalar);
	const float loss_sum = get_loss_scalar ? local_loss_sum() : 0.f;
	train_nerf_dp_backward(target_batch_size, counters[0], counters[1], get_loss_scalar, loss_sum);
	train_nerf_dp_end();
}

// ---- sample generation, possibly one step ahead on a second stream ----------------------------------------------------------
// The DDA march (generate_training_samples) is a serial, latency-bound kernel that keeps < 25 % of the SIMDs busy, and it depends
// only on the occupancy bitfield, the rng and rays_per_batch — not on the weights.  Once the counters of step n are known (after
// its loss kernel) the march of step n+1 is therefore launched on a second HIP stream and overlaps step n's forward / backward /
// optimizer.  Same inputs, same kernels, same results as the in-order schedule; skipped whenever an occupancy-grid update is due
// before step n+1 and discarded if any input changed in the meantime.
uint32_t Testbed::next_max_inference(uint32_t target_batch_size) const {  // testbed_nerf.cu:3185-3190
	const uint32_t max_samples = target_batch_size * 16;
	const NerfCounters& c = m_nerf.training.counters_rgb;
	if (c.measured_batch_size_before_compaction == 0) return max_samples;
	return next_multiple(std::min(c.measured_batch_size_before_compaction, max_samples), BATCH_SIZE_GRANULARITY);
}

void Testbed::launch_generate(void* stream, int slot, uint32_t R, uint32_t max_inference, const Pcg32& rng, bool next_to_backward) {
	NerfTraining& tr = m_nerf.training;
	const size_t r_cap = std::max<size_t>(R, 1u << 18);  // rays_per_batch is capped at 2^18 (2893): size once, no reallocation under a running step
	m_ray_indices.enlarge(r_cap * 4); m_rays.enlarge(r_cap * sizeof(NgpRay)); m_numsteps.enlarge(r_cap * 8);
	m_gen_counters.enlarge(32);
	uint32_t* counters = m_gen_counters.as<uint32_t>() + 4 * slot;  // [0] ray counter, [1] numsteps counter, [2] compacted numsteps counter (the step's)
	if (m_next_slot_zeroed == slot) m_next_slot_zeroed = -1;   // cleared by the previous step's post_words launch
	else HIP_CHECK_THROW(hipMemsetAsync(counters, 0, 8, (hipStream_t)stream));
	const int32_t dist_res[2] = {m_distortion.resolution[0], m_distortion.resolution[1]};
	const uint32_t n_rays_global = R * m_world_size, ray_offset = R * m_rank;
	NgpErrorMapCdf cdf_storage;
	profile_begin(PK_GEN_SAMPLES, stream);
	// run ahead on stream B the march shares the chip with the backward pass: the wave-per-ray kernel on two workgroups per CU (more of it costs the
	// backward pass more than it gains the march).  In stream order (first steps, the step after every occupancy update, a discarded prefetch)
	// nothing runs beside it: all workgroups at once.  With cone stepping the same two modes select the wave-per-ray kernel on the generated candidate sequence
	check(ngp_hip_generate_training_samples(stream, R, &m_aabb, max_inference, rng.state, rng.inc, counters + 0, counters + 1, m_ray_indices.as<uint32_t>(), m_rays.as<NgpRay>(),
	                                        m_numsteps.as<uint32_t>(), m_coords.as<NgpCoord>(), (uint32_t)tr.n_images_for_training, tr.dataset.metadata_gpu.as<NgpImageMeta>(),
	                                        tr.transforms_gpu.as<NgpXForm>(), m_nerf.density_grid_bitfield.as<uint8_t>(), m_max_level_rand_training, nullptr, tr.snap_to_pixel_centers, tr.train_envmap,
	                                        m_nerf.cone_angle_constant, m_distortion.params.as<float>() /* map->params(), 3241 */, dist_res, ray_offset, n_rays_global, tr.error_map_cdf(cdf_storage),
	                                        m_nerf.brick_summary_valid ? m_nerf.bitfield_brick_summary.as<uint32_t>() : nullptr,
	                                        next_to_backward ? NGP_MARCH_WAVE_PER_RAY_SHARED : NGP_MARCH_WAVE_PER_RAY), "generate_training_samples");
	profile_end(PK_GEN_SAMPLES, R, stream);
}

void Testbed::drop_prefetch() {
	if (m_prefetch.valid || m_grid_prefetch.valid) {
		HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream_b));
		m_prefetch.valid = false;
		m_grid_prefetch.valid = false;
	}
}

void Testbed::launch_grid_samples(void* stream, uint32_t n_uniform, uint32_t n_nonuniform) {
	NerfTraining& tr = m_nerf.training;
	const uint32_t n_elements = GRID_CELLS * (m_nerf.max_cascade + 1);
	float* grid = m_nerf.density_grid.as<float>();
	HIP_CHECK_THROW(hipMemsetAsync(m_grid_tmp.data(), 0, (size_t)n_elements * 4, (hipStream_t)stream));
	// (round 6) in Morton order of the cells: the same samples, neighbouring slots = neighbouring cells (density_grid.hip); `morton_grid_samples = false` is the reference's order
	const size_t ws_words = (size_t)ngp_hip_generate_grid_samples_morton_workspace_bytes() / 4;
	m_grid_sample_counters.enlarge(2 * ws_words * 4);   // one workspace per half: the second call must not overwrite what the first one's kernels still read
	uint32_t* ctr = m_grid_sample_counters.as<uint32_t>();
	if (m_morton_grid_samples)
		check(ngp_hip_generate_grid_samples_morton(stream, n_uniform, tr.density_grid_rng.state, tr.density_grid_rng.inc, m_nerf.density_grid_ema_step, &m_aabb, grid,
		                                           m_grid_positions.as<float>(), m_grid_indices.as<uint32_t>(), m_nerf.max_cascade + 1, -0.01f, ctr), "generate_grid_samples (uniform, Morton order)");
	else
		check(ngp_hip_generate_grid_samples_nonuniform(stream, n_uniform, tr.density_grid_rng.state, tr.density_grid_rng.inc, m_nerf.density_grid_ema_step, &m_aabb, grid,
		                                               m_grid_positions.as<float>(), m_grid_indices.as<uint32_t>(), m_nerf.max_cascade + 1, -0.01f), "generate_grid_samples (uniform)");
	tr.density_grid_rng.advance();
	if (m_morton_grid_samples)
		check(ngp_hip_generate_grid_samples_morton(stream, n_nonuniform, tr.density_grid_rng.state, tr.density_grid_rng.inc, m_nerf.density_grid_ema_step, &m_aabb, grid,
		                                           m_grid_positions.as<float>() + (size_t)n_uniform * 3, m_grid_indices.as<uint32_t>() + n_uniform, m_nerf.max_cascade + 1,
		                                           NERF_MIN_OPTICAL_THICKNESS, ctr + ws_words), "generate_grid_samples (nonuniform, Morton order)");
	else
		check(ngp_hip_generate_grid_samples_nonuniform(stream, n_nonuniform, tr.density_grid_rng.state, tr.density_grid_rng.inc, m_nerf.density_grid_ema_step, &m_aabb, grid,
		                                               m_grid_positions.as<float>() + (size_t)n_uniform * 3, m_grid_indices.as<uint32_t>() + n_uniform, m_nerf.max_cascade + 1,
		                                               NERF_MIN_OPTICAL_THICKNESS), "generate_grid_samples (nonuniform)");
	tr.density_grid_rng.advance();
}

// The step before an occupancy-grid update has no march to run ahead (the bitfield is about to change); stream B generates the update's sample positions instead.
// What the generators read — the density grid, its generator state, the ema step — does not change until the update itself; anything that could change it bumps
// m_state_version (or is compared below), and a prefetch that no longer matches is simply regenerated in stream order.
void Testbed::maybe_prefetch_grid_samples(uint32_t next_step) {
	NerfTraining& tr = m_nerf.training;
	if (next_step < 256 || tr.n_images_for_training == 0 || tr.n_images_for_training != tr.n_images_for_training_prev) return;   // full-grid phase / the update re-marks the grid first
	const uint32_t n_cascades = m_nerf.max_cascade + 1, n_elements = GRID_CELLS * n_cascades;
	const uint32_t n_uniform = GRID_CELLS / 4 * n_cascades, n_nonuniform = GRID_CELLS / 4 * n_cascades;
	const uint32_t n_samples = n_uniform + n_nonuniform;
	if (m_nerf.density_grid.bytes() != (size_t)n_elements * 4 || m_grid_positions.bytes() < (size_t)n_samples * 12 || m_grid_indices.bytes() < (size_t)n_samples * 4 ||
	    m_grid_tmp.bytes() < (size_t)n_elements * 4) return;   // buffers are sized by the first update
	PrefetchedGridSamples g;
	g.valid = true; g.step = next_step; g.n_uniform = n_uniform; g.n_nonuniform = n_nonuniform; g.ema_step = m_nerf.density_grid_ema_step;
	g.rng_state = tr.density_grid_rng.state; g.rng_inc = tr.density_grid_rng.inc; g.version = m_state_version; g.n_images = tr.n_images_for_training;
	const Pcg32 keep = tr.density_grid_rng;   // the update advances the generator itself
	launch_grid_samples(m_stream_b, n_uniform, n_nonuniform);
	tr.density_grid_rng = keep;
	if (!m_grid_prefetch_event) { hipEvent_t e; HIP_CHECK_THROW(hipEventCreateWithFlags(&e, STEP_EVENT_FLAGS)); m_grid_prefetch_event = e; }
	HIP_CHECK_THROW(hipEventRecord((hipEvent_t)m_grid_prefetch_event, (hipStream_t)m_stream_b));
	m_grid_prefetch = g;
}

bool Testbed::debug_grid_update_samples(bool regenerate, std::vector<float>& positions, std::vector<uint32_t>& indices, uint32_t& step) {
	const bool pending = m_grid_prefetch.valid;
	step = m_grid_prefetch.step;
	positions.clear(); indices.clear();
	if (!regenerate && !pending) return false;
	const uint32_t n_cascades = m_nerf.max_cascade + 1;
	const uint32_t n_uniform = GRID_CELLS / 4 * n_cascades, n = 2 * n_uniform;
	HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream_b));
	if (regenerate) {
		const Pcg32 keep = m_nerf.training.density_grid_rng;
		launch_grid_samples(m_stream, n_uniform, n_uniform);
		m_nerf.training.density_grid_rng = keep;
	}
	sync();
	positions.resize((size_t)n * 3); indices.resize(n);
	m_grid_positions.copy_to_host(positions.data(), (size_t)n * 12);
	m_grid_indices.copy_to_host(indices.data(), (size_t)n * 4);
	return pending;
}

void Testbed::maybe_prefetch_next(uint32_t target_batch_size) {
	if (!m_enable_prefetch || !m_train_continues) return;
	if (m_capture.valid && m_capture.step == m_training_step) return;   // a captured step keeps its scratch buffers until the test has read them
	const uint32_t next_step = m_training_step + 1;
	const uint32_t n_prep_to_skip = std::min(std::max(next_step / 16u, 1u), 16u);
	// stream B may only overwrite what stream A's queued work still reads once the host has seen the counters posted behind it — unless the step was begun without
	// waiting for them (a data-parallel host that reduces them in stream order): then stream B waits for the counters event itself.  Both prefetches need it: the
	// march overwrites the rays / coords the loss kernel read, the grid-sample generators overwrite the splat buffer, positions and indices of the previous update
	auto order_stream_b_behind_counters = [&]() {
		if (m_counters_host_seen) return;
		if (!m_counters_event_recorded) { HIP_CHECK_THROW(hipEventRecord((hipEvent_t)m_counters_event, (hipStream_t)m_stream)); m_counters_event_recorded = true; }
		HIP_CHECK_THROW(hipStreamWaitEvent((hipStream_t)m_stream_b, (hipEvent_t)m_counters_event, 0));
	};
	if (next_step % n_prep_to_skip == 0) { order_stream_b_behind_counters(); maybe_prefetch_grid_samples(next_step); return; }  // an occupancy-grid update (new bitfield) precedes that step: its sample positions go ahead instead
	if ((m_nerf.training.optimize_extrinsics || m_nerf.training.optimize_distortion) && m_nerf.training.n_steps_since_cam_update + 1 >= m_nerf.training.n_steps_between_cam_updates) return;  // new camera transforms / a new distortion map precede it (3060-3093)
	NerfCounters& c = m_nerf.training.counters_rgb;
	Pcg32 rng = m_rng;   // m_rng was already advanced for the next step (3380)
	PrefetchedSamples p;
	p.valid = true; p.step = next_step; p.R = c.rays_per_batch; p.max_inference = next_max_inference(target_batch_size); p.rng_state = rng.state;
	p.version = m_state_version; p.n_images = m_nerf.training.n_images_for_training; p.batch = target_batch_size; p.slot = m_gen_slot ^ 1;
	p.cdf_mode = m_nerf.training.cdf_mode();
	order_stream_b_behind_counters();
	// (holding the march back until the step's MLP backward kernel is through was measured twice: round 3 0.587 -> 0.593 ms; round 5 0.515 -> 0.548 ms on lego, fox unchanged — the march,
	// squeezed into a shorter window, costs the binning passes more than the MFMA kernel gains: profiles/r05_experiments.md section 8)
	// Data parallel over more than one rank: the gradient exchange (reduce-scatter, all-gather: RCCL kernels on a few workgroups, bound by the xGMI links) follows the
	// backward pass on stream A and leaves the chip idle for as long as the march takes on its own (~140 us vs 61-370 us of wire time at 8-2 ranks).  The
	// march is therefore held back until the gradients are final and runs BESIDE THE EXCHANGE with all its workgroups: the backward pass loses the neighbour that costs it
	// ~45 us (183 us alone, 229 beside the march), the march costs nothing.  Same kernel, same inputs, same samples.
	const bool behind_exchange = m_dp_comm && m_dp_march_behind_exchange && m_grid_grad_event_recorded;
	if (behind_exchange) HIP_CHECK_THROW(hipStreamWaitEvent((hipStream_t)m_stream_b, (hipEvent_t)m_grid_grad_event, 0));
	launch_generate(m_stream_b, p.slot, p.R, p.max_inference, rng, !behind_exchange);
	if (!m_prefetch_event) { hipEvent_t e; HIP_CHECK_THROW(hipEventCreateWithFlags(&e, STEP_EVENT_FLAGS)); m_prefetch_event = e; }
	HIP_CHECK_THROW(hipEventRecord((hipEvent_t)m_prefetch_event, (hipStream_t)m_stream_b));
	m_prefetch = p;
}

void Testbed::train_nerf_dp_begin(uint32_t target_batch_size, uint32_t counters_out[2], bool get_loss_scalar, bool wait_for_counters) {
	NerfTraining& tr = m_nerf.training;
	NerfCounters& c = tr.counters_rgb;
	if (target_batch_size % 256) throw std::runtime_error{"training batch size must be a multiple of 256"};
	const uint32_t max_samples = target_batch_size * 16;
	const uint32_t R = c.rays_per_batch;
	m_coords.enlarge((size_t)max_samples * sizeof(NgpCoord));
	m_mlp_out.enlarge((size_t)std::max(target_batch_size, max_samples) * OUT_STRIDE * 2);
	m_dloss.enlarge((size_t)target_batch_size * OUT_STRIDE * 2);
	m_coords_compacted.enlarge((size_t)target_batch_size * sizeof(NgpCoord));
	// (round 6) the compaction leaves an INDEX into the uncompacted batch's encoding rows instead of copying each kept sample's 64-byte row (128 B of traffic per sample in
	// the loss kernel, and the row roll-over); the backward pass reads row x_index[k] of m_x_all.  Base network family, rows from the network pass (no second forward).
	const bool use_x_index = m_x_row_index_mode && net_is_base_family() && !m_separate_forward;
	if (use_x_index) m_x_index.enlarge((size_t)target_batch_size * 4); else m_x_saved.enlarge((size_t)target_batch_size * 32 * 2);
	m_last_step_used_x_index = use_x_index;
	m_bwd_scratch.enlarge(ngp_hip_nerf_backward_scratch_bytes_for(&m_desc, target_batch_size));   // sized for this level table (778 -> 405 MB at 2^18 with base.json)
	m_enc_ws.enlarge(ngp_hip_nerf_encode_workspace_bytes(std::max(target_batch_size, next_max_inference(target_batch_size))));

	// prepare_for_training_steps (testbed_nerf.cu:2861-2868)
	// (the compacted-samples counter lives next to the march's counters and is cleared below; the loss kernel writes every one of the R loss
	// slots, zeros for the unused ones, so neither needs a memset on the step's critical chain)
	c.loss.enlarge(std::max<size_t>((size_t)R * m_world_size, 1u << 18) * 4);   // [n_rays_global]: the kernel's slot range
	// error map (re)allocation (2933-2939)
	if (tr.n_steps_since_error_map_update == 0 && !tr.dataset.metadata.empty()) {
		const uint32_t n_samples_per_image = (tr.n_steps_between_error_map_updates * R) / (uint32_t)tr.dataset.n_images;
		const int r = (int)(std::sqrt(std::sqrt((float)n_samples_per_image)) * 3.5f);
		tr.error_map_res[0] = std::min(r, tr.dataset.metadata[0].res[0]);
		tr.error_map_res[1] = std::min(r, tr.dataset.metadata[0].res[1]);
		tr.error_map_data.resize((size_t)tr.error_map_res[0] * tr.error_map_res[1] * tr.dataset.n_images * 4);
		tr.error_map_data.memset(0, m_stream);
	}

	// ---- train_nerf_step, first half (3138-3312): samples, inference on all of them, loss + compaction
	uint32_t max_inference;
	if (c.measured_batch_size_before_compaction == 0) {
		c.measured_batch_size_before_compaction = max_inference = max_samples;
	} else {
		max_inference = next_max_inference(target_batch_size);
	}
	if (m_training_step == 0) c.n_rays_total = 0;
	c.n_rays_total += R;
	tr.n_rays_since_error_map_update += R;

	const bool hit = m_prefetch.valid && m_prefetch.step == m_training_step && m_prefetch.R == R && m_prefetch.max_inference == max_inference && m_prefetch.rng_state == m_rng.state &&
	                 m_prefetch.version == m_state_version && m_prefetch.n_images == tr.n_images_for_training && m_prefetch.batch == target_batch_size &&
	                 m_prefetch.cdf_mode == tr.cdf_mode();
	if (hit) {
		HIP_CHECK_THROW(hipStreamWaitEvent((hipStream_t)m_stream, (hipEvent_t)m_prefetch_event, 0));
		m_gen_slot = m_prefetch.slot;
		m_prefetch.valid = false;
		++m_prefetch_hits;
	} else {
		drop_prefetch();
		launch_generate(m_stream, m_gen_slot, R, max_inference, m_rng, false);
	}
	uint32_t* gen_counters = m_gen_counters.as<uint32_t>() + 4 * m_gen_slot;
	uint32_t* compacted_counter = gen_counters + 2;
	const bool capturing = m_capture.armed;
	auto capture_copy = [&](DeviceBuffer& dst, const void* src, size_t bytes) {
		dst.resize(bytes);
		HIP_CHECK_THROW(hipMemcpyAsync(dst.data(), src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)m_stream));
	};
	if (capturing) {   // test hook (testbed.h StepCapture): what the march handed to the network pass, and the weights the step starts from
		StepCapture& cap = m_capture;
		cap.armed = false; cap.valid = true;
		cap.step = m_training_step; cap.R = R; cap.max_inference = max_inference; cap.n_rays_global = R * m_world_size; cap.ray_offset = R * m_rank;
		cap.target_batch_size = target_batch_size; cap.rng_state = m_rng.state; cap.rng_inc = m_rng.inc;
		capture_copy(cap.params, m_params.data(), m_n_params * 2);
		capture_copy(cap.ray_indices, m_ray_indices.data(), (size_t)R * 4);
		capture_copy(cap.rays, m_rays.data(), (size_t)R * sizeof(NgpRay));
		capture_copy(cap.numsteps, m_numsteps.data(), (size_t)R * 8);
		capture_copy(cap.coords, m_coords.data(), (size_t)max_inference * sizeof(NgpCoord));
		capture_copy(cap.gen_counters, gen_counters, 8);
		capture_copy(cap.density_grid_mean, m_nerf.density_grid_mean.data(), 4);
		capture_copy(cap.bitfield, m_nerf.density_grid_bitfield.data(), m_nerf.density_grid_bitfield.bytes());   // the occupancy bits as the step's first kernels see them
		cap.prefetch_hit = hit;
	}
	if (m_compact_slot_zeroed == m_gen_slot) m_compact_slot_zeroed = -1;   // cleared by the previous step's post_words launch
	else HIP_CHECK_THROW(hipMemsetAsync(compacted_counter, 0, 8, (hipStream_t)m_stream));   // word 2: the compaction counter, word 3: the ray queue of the forward pass
	const NgpNetDesc* desc = m_desc_gpu.as<NgpNetDesc>();
	const uint32_t n_rays_global = R * m_world_size;
	// inference over the (padded) pre-compaction samples with the TRAINING weights (3256).  The pass also stores every sample's encoding
	// (64 B): the loss kernel carries the rows of the kept samples through the compaction, which replaces the reference's second network
	// pass over the compacted batch (3330) — its only product that backward consumes is that encoding (ngp_hip.h "Forward pass").
	m_x_all.enlarge((size_t)max_inference * 32 * 2);
	profile_begin(PK_INFERENCE);
	{
		NgpNetVariant nv;
		const NgpNetVariant* variant = nullptr;
		if (!net_is_base_family()) {
			if (m_n_extra_dims) {   // every sample carries its image's extra dims (testbed_nerf.cu:1136, 1246): here as a row index per sample, expanded from the kept rays' runs
				NgpErrorMapCdf cdf_slots;
				const NgpErrorMapCdf* cdfp = tr.error_map_cdf(cdf_slots);
				m_ray_image.enlarge((size_t)R * 4); m_sample_slot_all.enlarge((size_t)max_inference * 4);
				HIP_CHECK_THROW(hipMemsetAsync(m_sample_slot_all.data(), 0, (size_t)max_inference * 4, (hipStream_t)m_stream));   // padding rows of the batch: any valid row
				check(ngp_hip_ray_images(m_stream, R, gen_counters + 0, m_ray_indices.as<uint32_t>(), n_rays_global, (uint32_t)tr.n_images_for_training, cdfp ? cdfp->cdf_img : nullptr, m_ray_image.as<uint32_t>()), "ray_images");
				check(ngp_hip_expand_ray_slots(m_stream, R, gen_counters + 0, m_ray_image.as<uint32_t>(), m_numsteps.as<uint32_t>(), max_inference, m_sample_slot_all.as<uint32_t>()), "expand_ray_slots");
			}
			variant = net_variant(nv, tr.extra_dims_gpu.as<float>(), m_n_extra_dims ? m_sample_slot_all.as<uint32_t>() : nullptr);
		}
		// which organisation: measured on this workload (testbed.h ENetworkPass).  The network variants' training pass has the fused organisation only
		// (ngp_hip_nerf_forward_ws hands them to it): nothing to choose there
		const bool two_kernel = !variant && tuner_pick(m_stream);
		if (two_kernel) check(ngp_hip_nerf_forward_ws(m_stream, desc, m_params.as<uint16_t>(), m_coords.as<float>(), 7, max_inference, m_mlp_out.as<uint16_t>(), OUT_STRIDE, m_x_all.as<uint16_t>(), m_enc_ws.data(), m_enc_ws.bytes(), nullptr), "nerf_inference (two kernels)");
		else check(ngp_hip_nerf_forward(m_stream, desc, m_params.as<uint16_t>(), m_coords.as<float>(), 7, max_inference, m_mlp_out.as<uint16_t>(), OUT_STRIDE, m_x_all.as<uint16_t>(), variant), "nerf_inference");
		tuner_done(m_stream);
	}
	profile_end(PK_INFERENCE, max_inference);
	if (tr.optimize_exposure) {
		if (m_world_size > 1 && !m_dp_comm) throw std::runtime_error{"optimize_exposure at world_size > 1 needs init_data_parallel (the exposure gradients are summed over the ranks)"};
		const size_t bytes = tr.dataset.n_images * 3 * sizeof(float);
		if (tr.cam_exposure_gradient_gpu.bytes() < bytes) { tr.cam_exposure_gradient_gpu.resize(bytes); tr.n_steps_since_cam_update = 0; }
		if (tr.n_steps_since_cam_update == 0) tr.cam_exposure_gradient_gpu.memset(0, m_stream);   // 2916-2919
	}
	if (tr.optimize_extrinsics) {
		if (m_world_size > 1 && !m_dp_comm) throw std::runtime_error{"optimize_extrinsics at world_size > 1 needs init_data_parallel (the camera gradients are summed over the ranks)"};
		const size_t bytes = tr.dataset.n_images * 3 * sizeof(float);
		if (tr.cam_pos_gradient_gpu.bytes() < bytes) { tr.cam_pos_gradient_gpu.resize(bytes); tr.cam_rot_gradient_gpu.resize(bytes); tr.n_steps_since_cam_update = 0; }
		if (tr.n_steps_since_cam_update == 0 || !tr.cam_gradient_window_open) { tr.cam_pos_gradient_gpu.memset(0, m_stream); tr.cam_rot_gradient_gpu.memset(0, m_stream); }   // 2916-2918
	}
	tr.cam_gradient_window_open = tr.optimize_extrinsics;
	if (tr.optimize_distortion) {
		if (m_world_size > 1 && !m_dp_comm) throw std::runtime_error{"optimize_distortion at world_size > 1 needs init_data_parallel (the map's gradients are summed over the ranks)"};
		if (tr.n_steps_since_cam_update == 0 || !tr.distortion_gradient_window_open) m_distortion.clear_gradients(m_stream, true);   // 2919-2920
	}
	tr.distortion_gradient_window_open = tr.optimize_distortion;
	const bool train_envmap = tr.train_envmap && m_envmap.n_params() > 0;
	if (train_envmap) {
		if (m_world_size > 1 && !m_dp_comm) throw std::runtime_error{"train_envmap at world_size > 1 needs init_data_parallel (the map's gradients are summed over the ranks)"};
		m_envmap.clear_gradients(m_stream, false);   // 2941-2944: every step
	}
	NgpLossExtras loss_extras{};
	if (tr.include_sharpness_in_error) {   // 2901-2912: the grid of the sharpest tile per cell, zeroed at step 0, decayed by 0.95 every step; 3303-3305
		tr.dataset.update_sharpness();
		const size_t n_cells = (size_t)GRID_CELLS * NGP_NERF_CASCADES;
		if (tr.sharpness_grid.bytes() < n_cells * 4) { tr.sharpness_grid.resize(n_cells * 4); tr.sharpness_grid.memset(0, m_stream); }
		if (m_training_step == 0) tr.sharpness_grid.memset(0, m_stream);
		else check(ngp_hip_decay_grid(m_stream, (uint32_t)n_cells, 0.95f, tr.sharpness_grid.as<float>()), "decay_sharpness_grid");
		loss_extras.sharpness_data = tr.dataset.sharpness_data.as<float>(); loss_extras.sharpness_grid = tr.sharpness_grid.as<float>();
		loss_extras.sharpness_res[0] = tr.dataset.sharpness_resolution[0]; loss_extras.sharpness_res[1] = tr.dataset.sharpness_resolution[1];
	}
	if (m_envmap.n_params() > 0) {   // 3219-3222: the training copy of the map, its gradient buffer only while it trains
		loss_extras.envmap_data = m_envmap.params.as<float>(); loss_extras.envmap_gradient = train_envmap ? m_envmap.gradients.as<float>() : nullptr;
		loss_extras.envmap_res[0] = m_envmap.resolution[0]; loss_extras.envmap_res[1] = m_envmap.resolution[1]; loss_extras.envmap_loss_type = (int)m_envmap.loss_type;
	}
	loss_extras.x_row_index_out = use_x_index ? m_x_index.as<uint32_t>() : nullptr;
	profile_begin(PK_LOSS);
	NgpErrorMapCdf cdf_storage;
	check(ngp_hip_compute_loss(m_stream, n_rays_global, &m_aabb, m_rng.state, m_rng.inc, target_batch_size, gen_counters + 0, LOSS_SCALE, OUT_STRIDE, m_background_color,
	                           (int)m_color_space, tr.random_bg_color, tr.linear_colors, (uint32_t)tr.n_images_for_training, tr.dataset.metadata_gpu.as<NgpImageMeta>(),
	                           m_mlp_out.as<uint16_t>(), compacted_counter, m_ray_indices.as<uint32_t>(), m_rays.as<NgpRay>(), m_numsteps.as<uint32_t>(),
	                           m_coords.as<NgpCoord>(), m_coords_compacted.as<NgpCoord>(), m_dloss.as<uint16_t>(), OUT_STRIDE, (int)tr.loss_type, c.loss.as<float>(),
	                           m_max_level_rand_training, nullptr, (int)m_nerf.rgb_activation, (int)m_nerf.density_activation, tr.snap_to_pixel_centers,
	                           tr.error_map_data.as<float>(), tr.error_map_res, m_nerf.density_grid_mean.as<float>(), tr.cam_exposure_gpu.as<float>(), tr.near_distance,
	                           tr.error_map_cdf(cdf_storage), use_x_index ? nullptr : m_x_all.as<uint16_t>(), use_x_index ? nullptr : m_x_saved.as<uint16_t>(), tr.depth_supervision_lambda, (int)tr.depth_loss_type,
	                           tr.optimize_exposure ? tr.cam_exposure_gradient_gpu.as<float>() : nullptr, &loss_extras), "compute_loss");
	profile_end(PK_LOSS, R);
	if (capturing) {   // the loss kernel's products before the roll-over pads and rescales them
		capture_copy(m_capture.numsteps_compacted, m_numsteps.data(), (size_t)R * 8);
		capture_copy(m_capture.coords_compacted, m_coords_compacted.data(), (size_t)target_batch_size * sizeof(NgpCoord));
		capture_copy(m_capture.dloss, m_dloss.data(), (size_t)target_batch_size * OUT_STRIDE * 2);
	}
	// NerfCounters::update_after_training reads the two counters (2870-2874) with blocking copies after the whole step.  They are
	// final once the loss kernel ran, so a one-wave kernel gathers them (and the loss sum when asked for) into pinned host memory
	// here and the host picks them up from an event: forward / backward are queued behind it without a gap, and the next step's
	// march can be launched (stream B) while they run.
	if (tr.optimize_extrinsics || tr.optimize_distortion) {
		// the camera-gradient kernel runs behind the backward pass and reads this step's rays; the next step's march (stream B, launched once the host has
		// seen the counters posted below) overwrites them — so they are set aside here, in stream order before the post (R x 36 bytes)
		m_cam_rays.enlarge((size_t)R * (sizeof(NgpRay) + 8)); m_cam_ray_indices.enlarge((size_t)R * 4);
		HIP_CHECK_THROW(hipMemcpyAsync(m_cam_rays.data(), m_rays.data(), (size_t)R * sizeof(NgpRay), hipMemcpyDeviceToDevice, (hipStream_t)m_stream));
		HIP_CHECK_THROW(hipMemcpyAsync((char*)m_cam_rays.data() + (size_t)R * sizeof(NgpRay), m_numsteps.data(), (size_t)R * 8, hipMemcpyDeviceToDevice, (hipStream_t)m_stream));
		HIP_CHECK_THROW(hipMemcpyAsync(m_cam_ray_indices.data(), m_ray_indices.data(), (size_t)R * 4, hipMemcpyDeviceToDevice, (hipStream_t)m_stream));
	}
	if (m_n_extra_dims) {
		// the compacted batch's extra-dims rows: the kept rays' compacted runs, then the roll-over rule for the padded tail; the compacted (count, base) pairs are set
		// aside for the latent-code gradient kernel behind the backward pass (the next step's march, launched once the host has seen the counters posted below,
		// overwrites m_numsteps) — all in stream order before the post
		m_sample_slot.enlarge((size_t)target_batch_size * 4); m_extra_numsteps.enlarge((size_t)R * 8);
		HIP_CHECK_THROW(hipMemsetAsync(m_sample_slot.data(), 0, (size_t)target_batch_size * 4, (hipStream_t)m_stream));
		check(ngp_hip_expand_ray_slots(m_stream, R, gen_counters + 0, m_ray_image.as<uint32_t>(), m_numsteps.as<uint32_t>(), target_batch_size, m_sample_slot.as<uint32_t>()), "expand_ray_slots (compacted)");
		check(ngp_hip_rollover_slots(m_stream, target_batch_size, compacted_counter, m_sample_slot.as<uint32_t>()), "rollover_slots");
		HIP_CHECK_THROW(hipMemcpyAsync(m_extra_numsteps.data(), m_numsteps.data(), (size_t)R * 8, hipMemcpyDeviceToDevice, (hipStream_t)m_stream));
	}
	const float* loss_sum_dev = nullptr;
	if (get_loss_scalar) {
		check(ngp_hip_reduce_sum_f32(m_stream, c.loss.as<float>(), R, m_loss_scalar_gpu.as<float>()), "reduce_sum");
		loss_sum_dev = m_loss_scalar_gpu.as<float>();
	}
	if (!m_host_words) {
		HIP_CHECK_THROW(hipHostMalloc(&m_host_words, 16, hipHostMallocMapped | hipHostMallocCoherent));
		hipEvent_t e; HIP_CHECK_THROW(hipEventCreateWithFlags(&e, STEP_EVENT_FLAGS)); m_counters_event = e;
	}
	// the same launch clears the other slot's march counters and compacted-samples counter: its last reader (the previous step's loss kernel) is long done, and the
	// march that will bump them is launched after the host has seen this step's counters
	m_post_tag = m_post_tag + 1 ? m_post_tag + 1 : 1;
	m_next_slot_zeroed = m_compact_slot_zeroed = m_gen_slot ^ 1;
	uint32_t* next_slot = m_gen_counters.as<uint32_t>() + 4 * (m_gen_slot ^ 1);   // its ray, sample and compaction counters + the forward pass's ray queue
	const bool compact_now = m_compact_backward && net_is_base_family() && !m_want_counters_event && !m_separate_forward && !(tr.optimize_extrinsics || tr.optimize_distortion);
	if (m_want_counters_event) {
		// somebody waits on the counters event (stream_wait_counters: a data-parallel host that reduces the counters in stream order): it is recorded right behind the post,
		// in front of the roll-overs, which are not needed for the counters
		check(ngp_hip_post_words(m_stream, gen_counters + 1, compacted_counter, (const uint32_t*)loss_sum_dev, m_post_tag, (uint32_t*)m_host_words, next_slot, 4, (double*)m_dp_counters_dev), "post_words");
		HIP_CHECK_THROW(hipEventRecord((hipEvent_t)m_counters_event, (hipStream_t)m_stream));
		check(ngp_hip_fill_rollover_training(m_stream, target_batch_size, compacted_counter, m_dloss.as<uint16_t>(), OUT_STRIDE, m_coords_compacted.as<float>(), 7,
		                                     use_x_index ? m_x_index.as<float>() : m_x_saved.as<float>(), use_x_index ? 1u : 16u), "fill_rollover");
	} else if (compact_now) {
		// the backward pass over the live samples: roll-overs, then the list of the samples with a non-zero loss gradient, THEN the post — the host launches the next step's march
		// when it sees it, and the backward pass's MFMA kernel (80 KiB of LDS per workgroup) must already be resident when that march arrives, or it runs at half its occupancy
		// beside it (measured: 47 -> 87 us with the post 20 us ahead of the kernel)
		check(ngp_hip_fill_rollover_training(m_stream, target_batch_size, compacted_counter, m_dloss.as<uint16_t>(), OUT_STRIDE, m_coords_compacted.as<float>(), 7, use_x_index ? m_x_index.as<float>() : m_x_saved.as<float>(), use_x_index ? 1u : 16u), "fill_rollover");
		m_coords_live.enlarge((size_t)target_batch_size * sizeof(NgpCoord)); m_live_index.enlarge((size_t)target_batch_size * 4);
		if (m_live_count.bytes() == 0) { m_live_count.resize(8); m_live_count.memset(0, m_stream); m_live_parity = 0; }
		check(ngp_hip_compact_live_samples(m_stream, target_batch_size, m_dloss.as<uint16_t>(), OUT_STRIDE, m_coords_compacted.as<float>(), 7, m_live_index.as<uint32_t>(), m_coords_live.as<float>(),
		                                   m_live_count.as<uint32_t>() + m_live_parity), "compact_live_samples");
		check(ngp_hip_post_words(m_stream, gen_counters + 1, compacted_counter, (const uint32_t*)loss_sum_dev, m_post_tag, (uint32_t*)m_host_words, next_slot, 4, (double*)m_dp_counters_dev), "post_words");
	} else {
		// the polling host needs no event (a record costs a few microseconds of dispatch gap): post and roll-overs in one launch, the post first
		check(ngp_hip_post_words_and_fill_rollover_training(m_stream, gen_counters + 1, compacted_counter, (const uint32_t*)loss_sum_dev, m_post_tag, (uint32_t*)m_host_words, next_slot, 4,
		                                                    (double*)m_dp_counters_dev, target_batch_size, compacted_counter, m_dloss.as<uint16_t>(), OUT_STRIDE, m_coords_compacted.as<float>(), 7,
		                                                    use_x_index ? m_x_index.as<float>() : m_x_saved.as<float>(), use_x_index ? 1u : 16u), "post_words + fill_rollover");
	}
	m_counters_event_recorded = m_want_counters_event;

	// ---- train_nerf_step, second half (3324-3332): backward on the compacted batch (gradients overwrite).  The reference's forward over
	// the compacted batch is the encoding that arrived with the compaction above; m_separate_forward restores the second pass (same bits).
	if (m_separate_forward) {
		profile_begin(PK_FORWARD);
		NgpNetVariant nvf;
		check(ngp_hip_nerf_forward(m_stream, desc, m_params.as<uint16_t>(), m_coords_compacted.as<float>(), 7, target_batch_size, m_mlp_out.as<uint16_t>(), OUT_STRIDE, m_x_saved.as<uint16_t>(),
		                           net_variant(nvf, tr.extra_dims_gpu.as<float>(), m_n_extra_dims ? m_sample_slot.as<uint32_t>() : nullptr)), "nerf_forward");
		profile_end(PK_FORWARD, target_batch_size);
	}
	profile_begin(PK_BACKWARD);
	if (!m_grid_grad_event) { hipEvent_t e; HIP_CHECK_THROW(hipEventCreateWithFlags(&e, STEP_EVENT_FLAGS)); m_grid_grad_event = e; }
	if (tr.optimize_extrinsics || tr.optimize_distortion) {
		// prepare_input_gradients (3327-3330): the backward pass also writes dL/d(pos, dir) of every compacted sample; compute_cam_gradient_train_nerf (3350-3378)
		// folds them into per-image position / rotation gradients.  The ray counter of this step's slot is stable until the step after next.
		m_coords_gradient.enlarge((size_t)target_batch_size * 6 * sizeof(float));
		NgpNetVariant nvc;
		const bool train_extra_dims_c = tr.dataset.n_extra_learnable_dims > 0 && tr.optimize_extra_dims;
		if (train_extra_dims_c) m_dl_dextra.enlarge((size_t)target_batch_size * m_n_extra_dims * 4);
		const NgpNetVariant* variant_c = net_variant(nvc, tr.extra_dims_gpu.as<float>(), m_n_extra_dims ? m_sample_slot.as<uint32_t>() : nullptr, train_extra_dims_c ? m_dl_dextra.as<float>() : nullptr);
		check(ngp_hip_nerf_backward(m_stream, desc, &m_desc, m_params.as<uint16_t>(), m_coords_compacted.as<float>(), 7, target_batch_size, use_x_index ? m_x_all.as<uint16_t>() : m_x_saved.as<uint16_t>(), m_dloss.as<uint16_t>(),
		                            OUT_STRIDE, m_grads.as<uint16_t>(), m_bwd_scratch.data(), m_bwd_scratch.bytes(), nullptr, nullptr, m_coords_gradient.as<float>(), variant_c, use_x_index ? m_x_index.as<uint32_t>() : nullptr), "nerf_backward (with input gradient)");
		if (m_want_grid_grad_event) HIP_CHECK_THROW(hipEventRecord((hipEvent_t)m_grid_grad_event, (hipStream_t)m_stream));
		profile_end(PK_BACKWARD, target_batch_size);
		check(ngp_hip_compute_cam_gradient(m_stream, n_rays_global, &m_aabb, m_rng.state, m_rng.inc, gen_counters + 0, tr.snap_to_pixel_centers,
		                                      tr.optimize_extrinsics ? tr.cam_pos_gradient_gpu.as<float>() : nullptr, tr.optimize_extrinsics ? tr.cam_rot_gradient_gpu.as<float>() : nullptr,
		                                      (uint32_t)tr.n_images_for_training, tr.dataset.metadata_gpu.as<NgpImageMeta>(), m_cam_ray_indices.as<uint32_t>(),
		                                      m_cam_rays.as<NgpRay>(), (const uint32_t*)((const char*)m_cam_rays.data() + (size_t)R * sizeof(NgpRay)), m_coords_compacted.as<NgpCoord>(),
		                                      m_coords_gradient.as<float>(), tr.error_map_cdf(cdf_storage), tr.transforms_gpu.as<NgpXForm>(),
		                                      tr.optimize_distortion ? m_distortion.gradients.as<float>() : nullptr, tr.optimize_distortion ? m_distortion.gradient_weights.as<float>() : nullptr,
		                                      m_distortion.resolution), "compute_cam_gradient");
		if (train_extra_dims_c) {   // as below: compute_extra_dims_gradient_train_nerf
			const size_t n = (size_t)m_n_extra_dims * (size_t)tr.n_images_for_training;
			tr.extra_dims_gradient_gpu.enlarge(n * 4);
			HIP_CHECK_THROW(hipMemsetAsync(tr.extra_dims_gradient_gpu.data(), 0, n * 4, (hipStream_t)m_stream));
			check(ngp_hip_extra_dims_gradient(m_stream, R, gen_counters + 0, m_ray_image.as<uint32_t>(), m_extra_numsteps.as<uint32_t>(), m_dl_dextra.as<float>(), m_n_extra_dims,
			                                  tr.extra_dims_gradient_gpu.as<float>()), "extra_dims_gradient");
		}
	} else {
		NgpNetVariant nv;
		const bool train_extra_dims = tr.dataset.n_extra_learnable_dims > 0 && tr.optimize_extra_dims;   // testbed_nerf.cu:2925
		if (train_extra_dims) m_dl_dextra.enlarge((size_t)target_batch_size * m_n_extra_dims * 4);
		const NgpNetVariant* variant = net_variant(nv, tr.extra_dims_gpu.as<float>(), m_n_extra_dims ? m_sample_slot.as<uint32_t>() : nullptr, train_extra_dims ? m_dl_dextra.as<float>() : nullptr);
		if (compact_now) {
			uint32_t* n_live = m_live_count.as<uint32_t>() + m_live_parity;
			check(ngp_hip_nerf_backward_live(m_stream, desc, &m_desc, m_params.as<uint16_t>(), m_coords_compacted.as<float>(), 7, target_batch_size, use_x_index ? m_x_all.as<uint16_t>() : m_x_saved.as<uint16_t>(), m_dloss.as<uint16_t>(),
			                                 OUT_STRIDE, m_grads.as<uint16_t>(), m_bwd_scratch.data(), m_bwd_scratch.bytes(), nullptr, m_want_grid_grad_event ? m_grid_grad_event : nullptr,
			                                 m_live_index.as<uint32_t>(), m_coords_live.as<float>(), n_live, m_live_count.as<uint32_t>() + (m_live_parity ^ 1u), use_x_index ? m_x_index.as<uint32_t>() : nullptr), "nerf_backward (live samples)");
			m_live_last_batch = target_batch_size; m_live_last_parity = m_live_parity;
			m_live_parity ^= 1u;
		} else
		check(ngp_hip_nerf_backward(m_stream, desc, &m_desc, m_params.as<uint16_t>(), m_coords_compacted.as<float>(), 7, target_batch_size, use_x_index ? m_x_all.as<uint16_t>() : m_x_saved.as<uint16_t>(), m_dloss.as<uint16_t>(),
		                            OUT_STRIDE, m_grads.as<uint16_t>(), m_bwd_scratch.data(), m_bwd_scratch.bytes(), nullptr,
		                            m_want_grid_grad_event ? m_grid_grad_event : nullptr, nullptr, variant, use_x_index ? m_x_index.as<uint32_t>() : nullptr), "nerf_backward");
		profile_end(PK_BACKWARD, target_batch_size);
		if (train_extra_dims) {   // compute_extra_dims_gradient_train_nerf (2925-2931, 3333-3346): per image, the sum over its rays' compacted samples
			const size_t n = (size_t)m_n_extra_dims * (size_t)tr.n_images_for_training;
			tr.extra_dims_gradient_gpu.enlarge(n * 4);
			HIP_CHECK_THROW(hipMemsetAsync(tr.extra_dims_gradient_gpu.data(), 0, n * 4, (hipStream_t)m_stream));
			check(ngp_hip_extra_dims_gradient(m_stream, R, gen_counters + 0, m_ray_image.as<uint32_t>(), m_extra_numsteps.as<uint32_t>(), m_dl_dextra.as<float>(), m_n_extra_dims,
			                                  tr.extra_dims_gradient_gpu.as<float>()), "extra_dims_gradient");
		}
	}
	m_grid_grad_event_recorded = m_want_grid_grad_event;
	m_rng.advance();  // 3380 (the generator and the loss kernel of this step both used the pre-advance state)

	m_counters_host_seen = wait_for_counters;
	if (!wait_for_counters) { counters_out[0] = counters_out[1] = 0; return; }   // the caller reduces m_dp_counters_dev in stream order instead
	// poll the host-mapped words (a couple of microseconds after the kernel's store; an event wake-up costs 10-20); the event is the fallback
	const volatile uint32_t* w = (const volatile uint32_t*)m_host_words;
	{
		const auto t0 = std::chrono::steady_clock::now();
		uint32_t spins = 0;
		while (__atomic_load_n((const uint32_t*)&w[3], __ATOMIC_ACQUIRE) != m_post_tag) {
			if ((++spins & 0x3ffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
				HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream));   // also surfaces a failed kernel instead of spinning forever
				break;
			}
		}
	}
	counters_out[0] = w[0]; counters_out[1] = w[1];
	uint32_t bits = w[2];
	memcpy(&m_local_loss_sum, &bits, 4);
}

void Testbed::stream_wait_counters(void* other_stream) {
	if (!m_counters_event) throw std::runtime_error{"stream_wait_counters: no step has been begun"};
	m_want_counters_event = true;   // from the next step on, recorded right behind the counters
	if (!m_counters_event_recorded) { HIP_CHECK_THROW(hipEventRecord((hipEvent_t)m_counters_event, (hipStream_t)m_stream)); m_counters_event_recorded = true; }   // this step: everything queued so far
	HIP_CHECK_THROW(hipStreamWaitEvent((hipStream_t)other_stream, (hipEvent_t)m_counters_event, 0));
}
void Testbed::stream_wait_grid_gradients(void* other_stream) {
	if (!m_grid_grad_event) throw std::runtime_error{"stream_wait_grid_gradients: no step has been begun"};
	m_want_grid_grad_event = true;
	if (!m_grid_grad_event_recorded) { HIP_CHECK_THROW(hipEventRecord((hipEvent_t)m_grid_grad_event, (hipStream_t)m_stream)); m_grid_grad_event_recorded = true; }
	HIP_CHECK_THROW(hipStreamWaitEvent((hipStream_t)other_stream, (hipEvent_t)m_grid_grad_event, 0));
}

// the encoding rows of the last step's (rolled-over) compacted batch, [batch][32] fp16 — gathered through the row index when the step left one instead of copies
std::vector<uint16_t> Testbed::debug_x_saved(size_t batch) {
	sync();
	std::vector<uint16_t> rows(batch * 32);
	if (!m_last_step_used_x_index) { if (batch) m_x_saved.copy_to_host(rows.data(), batch * 64); return rows; }
	std::vector<uint32_t> index(batch);
	std::vector<uint16_t> all(m_x_all.bytes() / 2);
	if (batch) m_x_index.copy_to_host(index.data(), batch * 4);
	if (!all.empty()) m_x_all.copy_to_host(all.data(), all.size() * 2);
	for (size_t k = 0; k < batch; ++k) {
		if ((size_t)index[k] * 32 + 32 > all.size()) throw std::runtime_error{"debug_x_saved: row index outside the uncompacted batch"};
		memcpy(&rows[k * 32], &all[(size_t)index[k] * 32], 64);
	}
	return rows;
}

const DeviceBuffer& Testbed::debug_buffer(const std::string& name) const {
	if (name == "mlp_out") return m_mlp_out;
	if (name == "coords_compacted") return m_coords_compacted;
	if (name == "dloss") return m_dloss;
	if (name == "x_saved") return m_x_saved;
	if (name == "grads") return m_grads;
	if (name == "coords") return m_coords;
	if (name == "coords_gradient") return m_coords_gradient;
	throw std::runtime_error{"debug_buffer: unknown buffer '" + name + "'"};
}

float Testbed::local_loss_sum() { return m_local_loss_sum; }  // of the step begun last, if it was begun with get_loss_scalar

void Testbed::train_nerf_dp_backward(uint32_t target_batch_size, uint32_t global_measured_before, uint32_t global_measured, bool get_loss_scalar, float global_loss_sum) {
	// counter feedback (it fixes the next step's rays_per_batch), then the next step's march on stream B
	update_after_training(target_batch_size, global_measured_before, global_measured, get_loss_scalar, global_loss_sum);
	m_train_continues = m_nerf.training.counters_rgb.measured_batch_size != 0;
	maybe_prefetch_next(target_batch_size);
	// the gradient all-reduce must be ordered on stream() after this point (and the optimizer after it): bench.py wraps the stream
	// as a torch ExternalStream; a caller without stream ordering calls sync() before and after its collective instead
}

// The data-parallel optimizer step (SURVEY.md §5 "Distributed communication backend"; round 5: fp16 on the wire, Ema sharded with the optimizer):
//   * gradient exchange: every rank sends slice q of its fp16 gradient vector to rank q (ngp_rccl_alltoall_f16: point to point over the node's direct xGMI links, no
//     arithmetic on the wire) and sums the world slices it receives IN RANK ORDER in fp32 with one fp16 rounding (ngp_hip_sum_slices_f16).  Each vector is the exact sum
//     of its rank's terms rounded once (§5.2); their fp32 sum rounded once is what the round-3 path (widen -> fp32 reduce-scatter -> narrow) produced, at 2 instead of
//     4 bytes per parameter on the wire and with a summation order that no longer depends on the library's ring (`dp_fp16_wire = False` restores that path);
//   * Adam AND Ema (36 B / parameter of state) run on this rank's shard only, one launch: 1 / world of the 62-85 us sweep (rounds 3-4 ran the 12 B / parameter Ema stage
//     over ALL parameters on every rank: 25 us per step that did not shrink with the ranks);
//   * the new fp16 TRAINING weights are all-gathered in place (2 B / parameter) — the next step's network passes read all of them.  The fp16 INFERENCE (Ema) weights
//     are read by the renderers and snapshots only: they stay sharded and are gathered when somebody needs them (dp_gather_inference_params: a collective; render()
//     runs it by itself when it is a collective anyway — render_sharded — and refuses otherwise).  The fp32 state (master weights, moments, Ema) stays sharded: rank r
//     owns elements [r * shard, (r + 1) * shard); dp_gather_optimizer_state makes it whole.
// Element for element the arithmetic of the replicated step: tests/test_dp_cpu.py (gloo, oracle; worlds of 2 and 3) and tests/test_dp_gpu.py (shards on one GPU) compare bit for bit.
void Testbed::optimizer_step_sharded() {
	join_side_ema();
	++m_optimizer_step;
	const uint32_t world = m_world_size, rank = m_rank;
	if (world > 1) m_dp_state_stale = true;
	const uint32_t shard = next_multiple(((uint32_t)m_n_params + world - 1) / world, 8u);
	if ((uint64_t)shard * world > m_n_params + DP_PARAM_SLACK) throw std::runtime_error{"optimizer_step_sharded: world size too large for the parameter buffers' slack"};
	const uint32_t off = shard * rank;
	const uint32_t mine = off < m_n_params ? std::min<uint32_t>(shard, (uint32_t)m_n_params - off) : 0u;
	const uint32_t mask = (m_train_network ? 1u : 0u) | (m_train_encoding ? 2u : 0u);
	const float ema_decay = m_use_ema ? m_ema_decay : 0.0f;
	profile_begin(PK_GRAD_EXCHANGE);
	if (m_dp_fp16_wire) {
		// (the gradient buffer carries DP_PARAM_SLACK elements behind the parameters: world x shard may exceed n_params by < 8 x world; what lies there is summed and never read)
		m_dp_grads_f32.enlarge((size_t)shard * world * 2);   // here: the world fp16 slices this rank receives
		check(ngp_rccl_alltoall_f16(m_dp_comm, m_stream, m_grads.as<uint16_t>(), m_dp_grads_f32.as<uint16_t>(), shard), "ngp_rccl_alltoall_f16 (gradients)");
		check(ngp_hip_sum_slices_f16(m_stream, world, shard, m_dp_grads_f32.as<uint16_t>(), m_grads.as<uint16_t>() + off), "sum_slices_f16 (gradient shard)");
	} else {
		m_dp_grads_f32.enlarge((size_t)shard * world * 4); m_dp_shard_f32.enlarge((size_t)shard * 4);
		check(ngp_hip_f16_to_f32(m_stream, (uint32_t)m_n_params, shard * world, m_grads.as<uint16_t>(), m_dp_grads_f32.as<float>()), "f16_to_f32 (gradients)");
		check(ngp_rccl_reduce_scatter_f32(m_dp_comm, m_stream, m_dp_grads_f32.as<float>(), m_dp_shard_f32.as<float>(), shard), "ngp_rccl_reduce_scatter_f32 (gradients)");
		check(ngp_hip_f32_to_f16(m_stream, shard, m_dp_shard_f32.as<float>(), m_grads.as<uint16_t>() + off), "f32_to_f16 (gradient shard)");
	}
	profile_end(PK_GRAD_EXCHANGE, m_n_params);
	profile_begin(PK_OPTIMIZER);
	const uint32_t matrix_here = m_n_matrix_params > off ? m_n_matrix_params - off : 0u;
	if (m_dp_sharded_ema) {
		if (mine) check(ngp_hip_optimizer_step(m_stream, mine, matrix_here, m_optimizer_step, m_learning_rate, m_beta1, m_beta2, m_epsilon, m_l2_reg, LOSS_SCALE, ema_decay,
		                                              m_grads.as<uint16_t>() + off, m_master.as<float>() + off, m_params.as<uint16_t>() + off, m_first_moments.as<float>() + off,
		                                              m_second_moments.as<float>() + off, m_ema.as<float>() + off, m_inference_params.as<uint16_t>() + off, mask), "optimizer_step (Adam + Ema, this rank's shard)");
		if (world > 1) m_dp_inference_stale = true;
	} else if (mine) {
		check(ngp_hip_optimizer_step(m_stream, mine, matrix_here, m_optimizer_step, m_learning_rate, m_beta1, m_beta2, m_epsilon, m_l2_reg, LOSS_SCALE, ema_decay,
		                                    m_grads.as<uint16_t>() + off, m_master.as<float>() + off, m_params.as<uint16_t>() + off, m_first_moments.as<float>() + off,
		                                    m_second_moments.as<float>() + off, nullptr, nullptr, mask | NGP_OPT_NO_EMA), "optimizer_step (Adam, this rank's shard)");
	}
	profile_end(PK_OPTIMIZER, mine);
	profile_begin(PK_PARAM_GATHER);
	check(ngp_rccl_allgather_f16(m_dp_comm, m_stream, m_params.as<uint16_t>(), shard), "ngp_rccl_allgather_f16 (weights)");
	profile_end(PK_PARAM_GATHER, m_n_params);
	if (!m_dp_sharded_ema)
		check(ngp_hip_optimizer_step(m_stream, (uint32_t)m_n_params, m_n_matrix_params, m_optimizer_step, m_learning_rate, m_beta1, m_beta2, m_epsilon, m_l2_reg, LOSS_SCALE, ema_decay, nullptr, nullptr,
		                                    m_params.as<uint16_t>(), nullptr, nullptr, m_ema.as<float>(), m_inference_params.as<uint16_t>(), NGP_OPT_EMA_ONLY), "optimizer_step (Ema, all parameters)");
	if (m_has_decay && m_optimizer_step >= m_decay_start && (m_decay_end == 0 || m_optimizer_step < m_decay_end) && m_decay_interval && m_optimizer_step % m_decay_interval == 0) {
		m_learning_rate *= m_decay_base;
	}
}

// The inference (Ema) fp16 weights after sharded steps: current inside this rank's shard only.  COLLECTIVE (every rank of the communicator): in-place all-gather, 2 B / parameter.
void Testbed::dp_gather_inference_params() {
	if (!m_dp_inference_stale) return;
	if (!m_dp_comm) { inference_params_from_training_weights("dp_gather_inference_params"); return; }
	const uint32_t shard = next_multiple(((uint32_t)m_n_params + m_world_size - 1) / m_world_size, 8u);
	check(ngp_rccl_allgather_f16(m_dp_comm, m_stream, m_inference_params.as<uint16_t>(), shard), "ngp_rccl_allgather_f16 (inference weights)");
	sync();
	m_dp_inference_stale = false;
}
// The communicator is gone (shutdown_data_parallel before a gather, a peer that died) and the Ema weights of the other ranks' shards with it.  What every rank still
// holds whole are the fp16 TRAINING weights (all-gathered at every step): a survivor renders and saves those — the model without the Ema smoothing — instead of
// being locked out until reset_network / load_snapshot.  Said once on stderr; the fp32 optimizer state stays marked stale (it cannot be rebuilt).
void Testbed::inference_params_from_training_weights(const char* who) {
	fprintf(stderr, "%s: the Ema (inference) weights were sharded over data-parallel ranks that are gone; falling back to the training weights of step %u (no Ema smoothing)\n", who, m_training_step);
	HIP_CHECK_THROW(hipMemcpyAsync(m_inference_params.data(), m_params.data(), (size_t)m_n_params * 2, hipMemcpyDeviceToDevice, (hipStream_t)m_stream));
	sync();
	m_dp_inference_stale = false;
}
// what a reader of the inference weights calls first: gathers when the call is a collective anyway (`collective`), refuses a stale copy otherwise
void Testbed::require_inference_params(const char* who, bool collective) {
	join_side_ema();
	if (!m_dp_inference_stale) return;
	if (collective && m_dp_comm) { dp_gather_inference_params(); return; }
	if (!m_dp_comm) { inference_params_from_training_weights(who); return; }
	throw std::runtime_error{std::string(who) + ": the inference (Ema) weights are sharded over the data-parallel ranks (dp_sharded_ema) and stale outside this rank's shard — call dp_gather_inference_params() on ALL ranks first (a collective), or set render_sharded on every rank so that render() is a collective and gathers by itself"};
}

void Testbed::optimizer_step() {  // Trainer::optimizer_step(stream, LOSS_SCALE) (testbed_nerf.cu:2950)
	if (m_dp_comm && m_dp_sharded_optimizer) { optimizer_step_sharded(); return; }
	if (m_dp_state_stale) throw std::runtime_error{"optimizer step on fp32 state that sharded data-parallel steps left stale outside this rank's shard: call dp_gather_optimizer_state() on all ranks before shutdown_data_parallel()"};
	++m_optimizer_step;
	const uint32_t mask = (m_train_network ? 1u : 0u) | (m_train_encoding ? 2u : 0u);
	const float ema_decay = m_use_ema ? m_ema_decay : 0.0f;
	// The Ema stage (12 of the step's 36 bytes per parameter: read the new fp16 weight and the fp32 average, write the average and the fp16 inference weight) feeds renderers
	// and snapshots, not the next training step: `ema_on_side_stream` runs it on stream B beside the next step's network pass instead of in front of it.  Element for element the
	// arithmetic of the one-launch step (NGP_OPT_NO_EMA / NGP_OPT_EMA_ONLY: the split of the data-parallel path, tests/test_dp_gpu.py).  OFF by default — measured
	// (profiles/r05_experiments.md section 7): the stage leaves the chain (lego 66 -> 46 us, fox 89 -> 68) but the pass it runs beside slows by 7-14 us and the two event pairs cost
	// the queues ~10 us: fox 0.644 -> 0.632 ms per step, lego 0.524 -> 0.528, image 0.242 -> 0.266, SDF 0.467 -> 0.502.
	const bool side_ema = m_ema_on_side_stream && m_stream_b && !(m_capture.valid && m_capture.step == m_training_step);
	profile_begin(PK_OPTIMIZER);
	if (side_ema) {
		join_side_ema();   // the previous Ema stage read the weights this Adam stage overwrites
		check(ngp_hip_optimizer_step(m_stream, (uint32_t)m_n_params, m_n_matrix_params, m_optimizer_step, m_learning_rate, m_beta1, m_beta2, m_epsilon, m_l2_reg, LOSS_SCALE, ema_decay,
		                                    m_grads.as<uint16_t>(), m_master.as<float>(), m_params.as<uint16_t>(), m_first_moments.as<float>(), m_second_moments.as<float>(), nullptr, nullptr,
		                                    mask | NGP_OPT_NO_EMA), "optimizer_step (Adam stage)");
	} else {
		join_side_ema();
		check(ngp_hip_optimizer_step(m_stream, (uint32_t)m_n_params, m_n_matrix_params, m_optimizer_step, m_learning_rate, m_beta1, m_beta2, m_epsilon, m_l2_reg, LOSS_SCALE, ema_decay,
		                                    m_grads.as<uint16_t>(), m_master.as<float>(), m_params.as<uint16_t>(), m_first_moments.as<float>(), m_second_moments.as<float>(), m_ema.as<float>(),
		                                    m_inference_params.as<uint16_t>(), mask), "optimizer_step");
	}
	profile_end(PK_OPTIMIZER, m_n_params);
	if (side_ema) {
		if (!m_adam_event) { hipEvent_t e; HIP_CHECK_THROW(hipEventCreateWithFlags(&e, STEP_EVENT_FLAGS)); m_adam_event = e; }
		if (!m_ema_event) { hipEvent_t e; HIP_CHECK_THROW(hipEventCreateWithFlags(&e, STEP_EVENT_FLAGS)); m_ema_event = e; }
		HIP_CHECK_THROW(hipEventRecord((hipEvent_t)m_adam_event, (hipStream_t)m_stream));
		HIP_CHECK_THROW(hipStreamWaitEvent((hipStream_t)m_stream_b, (hipEvent_t)m_adam_event, 0));
		check(ngp_hip_optimizer_step(m_stream_b, (uint32_t)m_n_params, m_n_matrix_params, m_optimizer_step, m_learning_rate, m_beta1, m_beta2, m_epsilon, m_l2_reg, LOSS_SCALE, ema_decay, nullptr, nullptr,
		                                    m_params.as<uint16_t>(), nullptr, nullptr, m_ema.as<float>(), m_inference_params.as<uint16_t>(), NGP_OPT_EMA_ONLY), "optimizer_step (Ema stage, stream B)");
		std::lock_guard<std::mutex> lock(m_ema_mutex);
		HIP_CHECK_THROW(hipEventRecord((hipEvent_t)m_ema_event, (hipStream_t)m_stream_b));
		m_ema_pending.store(true, std::memory_order_release);
	}
	// tcnn ExponentialDecay::step: after the nested step, lr *= decay_base whenever the step count hits start + k * interval
	if (m_has_decay && m_optimizer_step >= m_decay_start && (m_decay_end == 0 || m_optimizer_step < m_decay_end) && m_decay_interval && m_optimizer_step % m_decay_interval == 0) {
		m_learning_rate *= m_decay_base;
	}
}

void Testbed::train_nerf_dp_end() {
	NerfTraining& tr = m_nerf.training;
	optimizer_step();
	++m_training_step;
	if (tr.train_envmap && m_envmap.n_params() > 0) {   // 2954-2956
		if (m_dp_comm) check(ngp_rccl_allreduce_f32(m_dp_comm, m_stream, m_envmap.gradients.as<float>(), m_envmap.n_params()), "ngp_rccl_allreduce_f32 (envmap gradients)");
		m_envmap.optimizer_step(m_stream, LOSS_SCALE);
	}
	const bool zero_records = tr.counters_rgb.measured_batch_size == 0;
	if (zero_records) {
		m_loss_scalar = 0.f;
		fprintf(stderr, "Nerf training generated 0 samples. Aborting training.\n");
		m_train = false;
	}
	if (tr.dataset.n_extra_learnable_dims > 0 && tr.optimize_extra_dims && m_n_extra_dims) {   // 3029-3054: one host Adam per image on its latent code, every step
		const uint32_t n_img = (uint32_t)tr.n_images_for_training, ne = m_n_extra_dims;
		if (m_dp_comm) check(ngp_rccl_allreduce_f32(m_dp_comm, m_stream, tr.extra_dims_gradient_gpu.as<float>(), (uint64_t)n_img * ne), "ngp_rccl_allreduce_f32 (latent-code gradients)");
		std::vector<float> grad((size_t)n_img * ne);
		HIP_CHECK_THROW(hipMemcpyAsync(grad.data(), tr.extra_dims_gradient_gpu.data(), grad.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)m_stream));
		HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream));
		for (uint32_t i = 0; i < n_img; ++i) {
			NerfTraining::ExtraDimsAdam& a = tr.extra_dims_opt[i];
			// set_learning_rate(max(1e-3 * 0.33^(step / 128), lr / 1000)) with the optimizer's step count BEFORE this step (3043), then AdamOptimizer::step (adam_optimizer.h:38-45)
			const float lr0 = std::max(1e-3f * std::pow(0.33f, (float)(a.iter / 128)), m_learning_rate / 1000.0f);
			++a.iter;
			const float beta1 = 0.9f, beta2 = 0.99f, eps = 1e-8f;
			const float lr = lr0 * std::sqrt(1 - std::pow(beta2, (float)a.iter)) / (1 - std::pow(beta1, (float)a.iter));
			for (uint32_t j = 0; j < ne; ++j) {
				const float g = grad[(size_t)i * ne + j] / LOSS_SCALE;
				a.m[j] = beta1 * a.m[j] + (1 - beta1) * g;
				a.v[j] = beta2 * a.v[j] + (1 - beta2) * g * g;
				a.x[j] -= lr * a.m[j] / (std::sqrt(a.v[j]) + eps);
				grad[(size_t)i * ne + j] = a.x[j];   // "extra_dims_new_values" aliases the gradient array (3031)
			}
		}
		tr.extra_dims_gpu.copy_from_host(grad.data(), grad.size() * 4);
	}
	// camera-side trainables (3026, 3056-3135): only the per-image exposure is built
	tr.n_steps_since_cam_update += 1;
	const bool cam_update_due = tr.n_steps_since_cam_update >= tr.n_steps_between_cam_updates;
	if (tr.optimize_exposure && cam_update_due) {
		const uint32_t n_img = (uint32_t)tr.n_images_for_training;
		const float per_camera_loss_scale = (float)n_img / LOSS_SCALE / (float)tr.n_steps_between_cam_updates;
		std::vector<float> grad(tr.dataset.n_images * 3);
		if (m_dp_comm) check(ngp_rccl_allreduce_f32(m_dp_comm, m_stream, tr.cam_exposure_gradient_gpu.as<float>(), tr.dataset.n_images * 3), "ngp_rccl_allreduce_f32 (exposure gradients)");
		HIP_CHECK_THROW(hipMemcpyAsync(grad.data(), tr.cam_exposure_gradient_gpu.data(), grad.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)m_stream));
		HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream));
		if (tr.cam_exposure.size() < tr.dataset.n_images) tr.cam_exposure.resize(tr.dataset.n_images);
		float mean[3] = {0.f, 0.f, 0.f};
		for (uint32_t i = 0; i < n_img; ++i) {   // AdamOptimizer<Array3f>::step (adam_optimizer.h:38-45), lr = the trainer's current learning rate
			NerfTraining::ExposureAdam& a = tr.cam_exposure[i];
			++a.iter;
			const float beta1 = 0.9f, beta2 = 0.99f, eps = 1e-8f;
			const float lr = m_learning_rate * std::sqrt(1 - std::pow(beta2, (float)a.iter)) / (1 - std::pow(beta1, (float)a.iter));
			for (int c = 0; c < 3; ++c) {
				const float g = grad[i * 3 + c] * per_camera_loss_scale + a.x[c] * tr.exposure_l2_reg;
				a.m[c] = beta1 * a.m[c] + (1 - beta1) * g;
				a.v[c] = beta2 * a.v[c] + (1 - beta2) * g * g;
				a.x[c] -= lr * a.m[c] / (std::sqrt(a.v[c]) + eps);
				mean[c] += a.x[c];
			}
		}
		std::vector<float> exposures(tr.dataset.n_images * 3, 0.f);
		for (uint32_t i = 0; i < n_img; ++i) for (int c = 0; c < 3; ++c) { tr.cam_exposure[i].x[c] -= mean[c] / (float)n_img; exposures[i * 3 + c] = tr.cam_exposure[i].x[c]; }   // renormalise (3123-3129)
		tr.cam_exposure_gpu.copy_from_host(exposures.data(), exposures.size() * 4);
		++m_state_version;   // harmless for the march (it does not read exposures); keeps "inputs changed" bookkeeping honest
	}
	if (tr.optimize_extrinsics && cam_update_due) {   // 3063-3093
		const uint32_t n_img = (uint32_t)tr.n_images_for_training, n_all = (uint32_t)tr.dataset.n_images;
		const float per_camera_loss_scale = (float)n_img / LOSS_SCALE / (float)tr.n_steps_between_cam_updates;
		if (m_dp_comm) {
			check(ngp_rccl_allreduce_f32(m_dp_comm, m_stream, tr.cam_pos_gradient_gpu.as<float>(), n_all * 3), "ngp_rccl_allreduce_f32 (camera position gradients)");
			check(ngp_rccl_allreduce_f32(m_dp_comm, m_stream, tr.cam_rot_gradient_gpu.as<float>(), n_all * 3), "ngp_rccl_allreduce_f32 (camera rotation gradients)");
		}
		tr.cam_pos_gradient.resize(n_all * 3); tr.cam_rot_gradient.resize(n_all * 3);
		HIP_CHECK_THROW(hipMemcpyAsync(tr.cam_pos_gradient.data(), tr.cam_pos_gradient_gpu.data(), n_all * 12, hipMemcpyDeviceToHost, (hipStream_t)m_stream));
		HIP_CHECK_THROW(hipMemcpyAsync(tr.cam_rot_gradient.data(), tr.cam_rot_gradient_gpu.data(), n_all * 12, hipMemcpyDeviceToHost, (hipStream_t)m_stream));
		HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream));
		if (tr.cam_pos_offset.size() < n_all) tr.cam_pos_offset.resize(n_all);
		if (tr.cam_rot_offset.size() < n_all) tr.cam_rot_offset.resize(n_all);
		for (uint32_t i = 0; i < n_img; ++i) {
			float pos_gradient[3], rot_gradient[3];
			for (int c = 0; c < 3; ++c) {
				pos_gradient[c] = tr.cam_pos_gradient[i * 3 + c] * per_camera_loss_scale;
				rot_gradient[c] = tr.cam_rot_gradient[i * 3 + c] * per_camera_loss_scale;
				pos_gradient[c] += tr.cam_pos_offset[i].variable[c] * tr.extrinsic_l2_reg;
				rot_gradient[c] += tr.cam_rot_offset[i].variable[c] * tr.extrinsic_l2_reg;
			}
			// decays by a third every 128 camera updates, floored at a thousandth of the trainer's current learning rate (3076-3077)
			tr.cam_pos_offset[i].h.learning_rate = std::max(tr.extrinsic_learning_rate * std::pow(0.33f, (float)(tr.cam_pos_offset[i].iter / 128)), m_learning_rate / 1000.0f);
			tr.cam_rot_offset[i].h.learning_rate = std::max(tr.extrinsic_learning_rate * std::pow(0.33f, (float)(tr.cam_rot_offset[i].iter / 128)), m_learning_rate / 1000.0f);
			tr.cam_pos_offset[i].step(pos_gradient);
			tr.cam_rot_offset[i].step(rot_gradient);
		}
		drop_prefetch();            // no march is in flight (maybe_prefetch_next skipped this step); a stale one would be discarded through the version below
		tr.update_transforms();
		++m_state_version;
	}
	if (tr.optimize_distortion && cam_update_due) {   // 3085-3092
		if (m_dp_comm) {
			check(ngp_rccl_allreduce_f32(m_dp_comm, m_stream, m_distortion.gradients.as<float>(), m_distortion.n_params()), "ngp_rccl_allreduce_f32 (distortion gradients)");
			check(ngp_rccl_allreduce_f32(m_dp_comm, m_stream, m_distortion.gradient_weights.as<float>(), m_distortion.n_params()), "ngp_rccl_allreduce_f32 (distortion gradient weights)");
		}
		check(ngp_hip_safe_divide(m_stream, (uint32_t)m_distortion.n_params(), m_distortion.gradients.as<float>(), m_distortion.gradient_weights.as<float>()), "safe_divide");
		m_distortion.optimizer_step(m_stream, LOSS_SCALE * (float)tr.n_steps_between_cam_updates);
		drop_prefetch();
		++m_state_version;   // the ray generator reads the map
	}
	if (cam_update_due) tr.n_steps_since_cam_update = 0;   // 3134 (train_camera false: the reference never resets; the window only matters when training)
	// error map -> CDFs (2971-3023): low-overhead enough to be always on in the reference; sampling from them is a separate switch
	tr.n_steps_since_error_map_update += 1;
	if (tr.n_steps_since_error_map_update >= tr.n_steps_between_error_map_updates && tr.error_map_res[0] > 0 && tr.dataset.n_images > 0) {
		const uint32_t n_img = (uint32_t)tr.dataset.n_images, W = (uint32_t)tr.error_map_res[0], H = (uint32_t)tr.error_map_res[1];
		tr.cdf_res[0] = tr.error_map_res[0]; tr.cdf_res[1] = tr.error_map_res[1];
		tr.cdf_x_cond_y.resize((size_t)W * H * n_img * 4); tr.cdf_y.resize((size_t)H * n_img * 4); tr.cdf_img.resize((size_t)n_img * 4);
		// data parallel: every rank deposited the error of its own rays; the CDFs are built from the sum, identical on every rank
		if (m_dp_comm) check(ngp_rccl_allreduce_f32(m_dp_comm, m_stream, tr.error_map_data.as<float>(), (uint64_t)W * H * n_img), "ngp_rccl_allreduce_f32 (error map)");
		check(ngp_hip_construct_cdf_2d(m_stream, n_img, H, W, tr.error_map_data.as<float>(), tr.cdf_x_cond_y.as<float>(), tr.cdf_y.as<float>()), "construct_cdf_2d");
		check(ngp_hip_construct_cdf_1d(m_stream, n_img, H, tr.cdf_y.as<float>(), tr.cdf_img.as<float>()), "construct_cdf_1d");
		// image CDF on the CPU ("single-threaded anyway", 2999-3015)
		tr.pmf_img_cpu.resize(n_img);
		HIP_CHECK_THROW(hipMemcpyAsync(tr.pmf_img_cpu.data(), tr.cdf_img.data(), (size_t)n_img * 4, hipMemcpyDeviceToHost, (hipStream_t)m_stream));
		HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream));
		std::vector<float> cdf_img_cpu = tr.pmf_img_cpu;
		float cum = 0;
		for (float& f : cdf_img_cpu) { cum += f; f = cum; }
		const float norm = 1.0f / cum;
		for (size_t i = 0; i < cdf_img_cpu.size(); ++i) {
			constexpr float MIN_PMF = 0.1f;
			tr.pmf_img_cpu[i] = (1.0f - MIN_PMF) * tr.pmf_img_cpu[i] * norm + MIN_PMF / (float)n_img;
			cdf_img_cpu[i] = (1.0f - MIN_PMF) * cdf_img_cpu[i] * norm + MIN_PMF * (float)(i + 1) / (float)n_img;
		}
		HIP_CHECK_THROW(hipMemcpyAsync(tr.cdf_img.data(), cdf_img_cpu.data(), (size_t)n_img * 4, hipMemcpyHostToDevice, (hipStream_t)m_stream));
		HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream));   // cdf_img_cpu goes out of scope
		tr.n_steps_since_error_map_update = 0;
		tr.n_rays_since_error_map_update = 0;
		tr.is_cdf_valid = true;
		tr.n_steps_between_error_map_updates = (uint32_t)(tr.n_steps_between_error_map_updates * 1.5f);
		if (tr.cdf_mode()) ++m_state_version;   // a march that ran ahead used the previous CDFs
	}
}

const NgpErrorMapCdf* NerfTraining::error_map_cdf(NgpErrorMapCdf& storage) const {
	const uint32_t mode = cdf_mode();
	if (!mode) return nullptr;
	storage.cdf_x_cond_y = (mode & 1u) ? cdf_x_cond_y.as<float>() : nullptr;
	storage.cdf_y = (mode & 1u) ? cdf_y.as<float>() : nullptr;
	storage.cdf_img = (mode & 2u) ? cdf_img.as<float>() : nullptr;
	storage.res[0] = cdf_res[0]; storage.res[1] = cdf_res[1];
	return &storage;
}

void Testbed::update_after_training(uint32_t target_batch_size, uint32_t counter, uint32_t compacted_counter, bool get_loss_scalar, float loss_sum) {  // 2870-2894
	NerfCounters& c = m_nerf.training.counters_rgb;
	c.measured_batch_size = 0;
	c.measured_batch_size_before_compaction = 0;
	if (counter == 0 || compacted_counter == 0) return;
	// with W ranks the counters are global sums; every rank derives the same per-rank figures
	c.measured_batch_size_before_compaction = counter / m_world_size;
	c.measured_batch_size = compacted_counter / m_world_size;
	if (get_loss_scalar) m_loss_scalar = loss_sum * (float)c.measured_batch_size / (float)target_batch_size;
	c.rays_per_batch = (uint32_t)((float)c.rays_per_batch * (float)target_batch_size / (float)c.measured_batch_size);
	c.rays_per_batch = std::min(next_multiple(c.rays_per_batch, BATCH_SIZE_GRANULARITY), 1u << 18);
}

// ---- rendering --------------------------------------------------------------------------------------------------
// The frame leaves the device through ONE pinned staging buffer that lives as long as the Testbed (download()), never by a hipMemcpy into the caller's pageable
// array: the runtime pins such a destination for the DMA, and when the array is freed — a 1080 x 1920 frame is 33 MB, above glibc's largest mmap threshold, so every
// frame's array is a mapping of its own — the unmap of pages the GPU had mapped stalls the process's queues for 18-25 ms.  That, not the tracer, was the fox leg's
// "38 MP/s" of round 4 (profiles/r05_fox_render_timeline.txt: 20 ms of kernels per frame, 25-30 ms of idle GPU between two frames; 105 MP/s with the staging buffer).
// One staging buffer, two possible callers at once (render() on the caller's thread, the worker thread of request_nerf_render_async): the whole copy-out is one critical
// section, so a larger frame on one thread never frees the buffer the other is reading.  Only the training stream is drained — the copy is queued on it;
// joining the side-stream Ema stage is the business of whoever reads the inference weights, not of a frame copy.
void Testbed::download(const void* device_src, size_t bytes, void* host_dst) {
	std::lock_guard<std::mutex> lock(m_download_mutex);
	if (m_pinned_bytes < bytes) {
		if (m_pinned) (void)hipHostFree(m_pinned);
		m_pinned = nullptr; m_pinned_bytes = 0;
		HIP_CHECK_THROW(hipHostMalloc(&m_pinned, bytes, hipHostMallocDefault));
		m_pinned_bytes = bytes;
	}
	HIP_CHECK_THROW(hipMemcpyAsync(m_pinned, device_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)m_stream));
	HIP_CHECK_THROW(hipStreamSynchronize((hipStream_t)m_stream));
	memcpy(host_dst, m_pinned, bytes);
}

void Testbed::render_to_cpu(int width, int height, int spp, bool linear, float* out) {  // python_api.cu:132-190 (no camera path / motion blur); out: height * width * 4 floats
	if (m_n_params == 0) throw std::runtime_error{"render(): no network"};
	require_inference_params("render()", render_is_collective());
	RenderBuffer& rb = m_windowless_render_surface;
	rb.resize(width, height);
	rb.reset_accumulation();
	m_render_samples_evaluated = 0;
	if (m_autofocus) autofocus();   // python_api.cu:174-176
	auto start = std::chrono::steady_clock::now();
	const float no_rolling_shutter[4] = {0.f, 0.f, 0.f, 0.f};   // python_api.cu:177 (Vector4f::Zero())
	for (int i = 0; i < spp; ++i) render_frame(m_camera, m_camera, no_rolling_shutter, rb, !linear);
	const auto traced = std::chrono::steady_clock::now();
	fetch_render_surface(rb, out);
	m_stats.render_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - start).count();
	if (m_render_trace) fprintf(stderr, "render: frames %.3f ms, device -> host %.3f ms\n", std::chrono::duration<float, std::milli>(traced - start).count(),
	                            std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - traced).count());
}
std::vector<float> Testbed::render_to_cpu(int width, int height, int spp, bool linear) {
	std::vector<float> out((size_t)width * height * 4);
	render_to_cpu(width, height, spp, linear, out.data());
	return out;
}

void Testbed::render_with_rolling_shutter_to_cpu(const Mat34& camera_transform_start, const Mat34& camera_transform_end, const float rolling_shutter[4], int width, int height,
                                                 int spp, bool linear, float* out) {  // python_api.cu:262-275
	if (m_n_params == 0) throw std::runtime_error{"render_with_rolling_shutter(): no network"};
	require_inference_params("render_with_rolling_shutter()", render_is_collective());
	RenderBuffer& rb = m_windowless_render_surface;
	rb.resize(width, height);
	rb.reset_accumulation();
	m_render_samples_evaluated = 0;
	const Mat34 c0 = m_nerf.training.dataset.nerf_matrix_to_ngp(camera_transform_start), c1 = m_nerf.training.dataset.nerf_matrix_to_ngp(camera_transform_end);
	auto start = std::chrono::steady_clock::now();
	for (int i = 0; i < spp; ++i) {
		if (m_autofocus) autofocus();
		render_frame(c0, c1, rolling_shutter, rb, !linear);
	}
	fetch_render_surface(rb, out);
	m_stats.render_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - start).count();
}
std::vector<float> Testbed::render_with_rolling_shutter_to_cpu(const Mat34& camera_transform_start, const Mat34& camera_transform_end, const float rolling_shutter[4], int width, int height,
                                                               int spp, bool linear) {
	std::vector<float> out((size_t)width * height * 4);
	render_with_rolling_shutter_to_cpu(camera_transform_start, camera_transform_end, rolling_shutter, width, height, spp, linear, out.data());
	return out;
}

// ---- a frame rendered by several ranks -------------------------------------------------------------------------------------
// Rows [r * ceil(H / P), (r + 1) * ceil(H / P)) go to rank r (the last rank's range is cut at H; equal chunks keep the gather one collective).  OPT-IN: with
// `render_sharded` set on every rank of a data-parallel communicator (init_data_parallel) render() is a COLLECTIVE — sharded over the ranks, every rank returns the
// whole frame.  Without the flag (the default) render() is local and traces the whole frame, so one rank alone may render a screenshot or an eval frame during or
// after data-parallel training.  set_render_shard is the same split without a communicator: the caller renders (rank, world) and owns the gather (tests; a host with its own transport).
void Testbed::set_render_shard(uint32_t rank, uint32_t world) {
	if (world == 0 || rank >= world) throw std::runtime_error{"set_render_shard: bad rank / world"};
	m_render_shard_rank = rank; m_render_shard_world = world;
}
void Testbed::render_shard_rows(int height, int& row_begin, int& row_end) const {
	const bool dp = render_is_collective();
	const uint32_t world = dp ? m_world_size : m_render_shard_world, rank = dp ? m_rank : m_render_shard_rank;
	const int rows_per = (height + (int)world - 1) / (int)world;
	row_begin = std::min(height, (int)rank * rows_per);
	row_end = std::min(height, row_begin + rows_per);
}
void Testbed::fetch_render_surface(RenderBuffer& rb, float* out) {
	const int W = rb.res[0], H = rb.res[1];
	const size_t frame_bytes = (size_t)W * H * 16;
	if (!render_is_collective()) { download(rb.surface.data(), frame_bytes, out); return; }
	// all-gather of the ranks' row ranges (RCCL over xGMI, in place in a buffer of world equal chunks), then one copy to the host
	int row_begin, row_end;
	render_shard_rows(H, row_begin, row_end);
	const int rows_per = (H + (int)m_world_size - 1) / (int)m_world_size;
	const size_t chunk_floats = (size_t)rows_per * W * 4;
	m_render_gather.enlarge(chunk_floats * m_world_size * 4);
	float* gather = m_render_gather.as<float>();
	if (row_end > row_begin) HIP_CHECK_THROW(hipMemcpyAsync(gather + chunk_floats * m_rank, rb.surface.as<float>() + (size_t)row_begin * W * 4, (size_t)(row_end - row_begin) * W * 16, hipMemcpyDeviceToDevice, (hipStream_t)m_stream));
	check(ngp_rccl_allgather_f32(m_dp_comm, m_stream, gather, chunk_floats), "ngp_rccl_allgather_f32 (frame rows)");
	download(gather, frame_bytes, out);
}

void Testbed::autofocus() {  // testbed.cu:2933-2941: focus on m_autofocus_target
	const float* c = m_camera.m;   // column-major 3x4: view dir = column 2, position = column 3
	const float d = c[6] * (m_autofocus_target[0] - c[9]) + c[7] * (m_autofocus_target[1] - c[10]) + c[8] * (m_autofocus_target[2] - c[11]);
	const float new_slice_plane_z = std::max(d, 0.1f) - m_scale;
	if (new_slice_plane_z != m_slice_plane_z) {
		m_slice_plane_z = new_slice_plane_z;
		if (m_aperture_size != 0.0f) m_windowless_render_surface.reset_accumulation();
	}
}

void Testbed::render_frame(const Mat34& cam0, const Mat34& cam1, const float rolling_shutter[4], RenderBuffer& rb, bool to_srgb) {  // testbed.cu:2695-2911
	rb.frame_buffer.memset(0, m_stream);
	rb.depth_buffer.memset(0, m_stream);
	if (m_testbed_mode == ETestbedMode::Image) {
		render_image(rb);
		if (rb.spp == 0) rb.accumulate_buffer.memset(0, m_stream);
		rb.color_space = m_color_space;
		check(ngp_hip_accumulate(m_stream, rb.res, rb.frame_buffer.as<float>(), rb.accumulate_buffer.as<float>(), (float)rb.spp, (int)rb.color_space), "accumulate");
		++rb.spp;
		check(ngp_hip_tonemap(m_stream, rb.res, m_exposure, m_background_color, rb.accumulate_buffer.as<float>(), (int)rb.color_space, to_srgb ? NGP_COLOR_SRGB : NGP_COLOR_LINEAR,
		                      (int)rb.tonemap_curve, 0, rb.surface.as<float>()), "tonemap");
		sync();
		return;
	}
	if (m_testbed_mode == ETestbedMode::Sdf) throw std::runtime_error{"rendering an SDF (sphere tracing, testbed_sdf.cu) is outside the scope of this build; P2 covers the training step"};
	const float focal_length[2] = {m_relative_focal_length[0] * (float)rb.res[m_fov_axis] * m_zoom, m_relative_focal_length[1] * (float)rb.res[m_fov_axis] * m_zoom};
	const float screen_center[2] = {(0.5f - m_screen_center[0]) * m_zoom + 0.5f, (0.5f - m_screen_center[1]) * m_zoom + 0.5f};
	render_nerf(rb, focal_length, cam0, cam1, rolling_shutter, screen_center);
	// CudaRenderBuffer::accumulate / tonemap (render_buffer.cu:609-664)
	if (rb.spp == 0) rb.accumulate_buffer.memset(0, m_stream);
	rb.color_space = m_color_space;
	check(ngp_hip_accumulate(m_stream, rb.res, rb.frame_buffer.as<float>(), rb.accumulate_buffer.as<float>(), (float)rb.spp, (int)rb.color_space), "accumulate");
	++rb.spp;
	check(ngp_hip_tonemap(m_stream, rb.res, m_exposure, m_background_color, rb.accumulate_buffer.as<float>(), (int)rb.color_space, to_srgb ? NGP_COLOR_SRGB : NGP_COLOR_LINEAR,
	                      (int)rb.tonemap_curve, 0, rb.surface.as<float>()), "tonemap");
	sync();
}

void Testbed::prepare_nerf_masks() {  // testbed_nerf.cu:2339-2352
	m_n_render_masks = (uint32_t)m_render_masks.size();
	if (m_render_masks.empty()) return;
	const NgpMask3D first_mask = m_render_masks[0];
	if (first_mask.shape != (int)EMaskShape::All) {   // an `All` mask of the opposite mode goes in front (and stays there, like the reference's member vector)
		m_render_masks.insert(m_render_masks.begin(), Mask3D::All(first_mask.mode == (int)EMaskMode::Add ? EMaskMode::Subtract : EMaskMode::Add).pod);
		m_n_render_masks = (uint32_t)m_render_masks.size();
	}
	m_render_masks_gpu.resize(m_render_masks.size() * sizeof(NgpMask3D));
	m_render_masks_gpu.copy_from_host(m_render_masks.data(), m_render_masks.size() * sizeof(NgpMask3D));
}

void Testbed::render_nerf(RenderBuffer& rb, const float focal_length[2], const Mat34& cam0, const Mat34& cam1, const float rolling_shutter[4], const float screen_center[2]) {  // testbed_nerf.cu:2354-2500, 2047-2267
	// A frame rendered by several ranks (SURVEY.md §8e "Render: image tiles/rows per rank + gather"): this rank traces the rows [row_begin, row_end) only.  Rays are
	// independent and every per-pixel random number is keyed by the pixel's index in the whole frame, so the rows are the rows of the frame rendered at once;
	// render_to_cpu gathers them.  n_pixels below = the pixels THIS rank traces.
	int row_begin = 0, row_end = rb.res[1];
	render_shard_rows(rb.res[1], row_begin, row_end);
	const uint32_t n_pixels = (uint32_t)rb.res[0] * (uint32_t)(row_end - row_begin);
	if (n_pixels == 0) return;
	const size_t n_el = next_multiple(n_pixels, BATCH_SIZE_GRANULARITY) + 256;   // + 256: the input-gradient pass of the Normals mode works on multiples of 256
	for (int b = 0; b < 2; ++b) { m_tr_payload[b].enlarge(n_el * sizeof(NgpPayload)); m_tr_rgba[b].enlarge(n_el * 16); m_tr_depth[b].enlarge(n_el * 4); }
	m_tr_hit_payload.enlarge(n_el * sizeof(NgpPayload)); m_tr_hit_rgba.enlarge(n_el * 16); m_tr_hit_depth.enlarge(n_el * 4);
	m_tr_net_in.enlarge(n_el * 8 * sizeof(NgpCoord)); m_tr_net_out.enlarge(n_el * 8 * OUT_STRIDE * 2);
	m_tr_counters.enlarge((2 + 8 + 2) * 4);   // [1] finished rays, [2..9] alive rays per range, [10] NgpCompactOut::blocks_done
	uint32_t* hit_counter = m_tr_counters.as<uint32_t>() + 1;

	const int lens_mode = m_nerf.render_with_lens_distortion ? m_nerf.render_lens_proxy.lens_mode : 0;   // testbed_nerf.cu:2381
	const uint32_t sample_index = rb.spp;
	float plane_z = m_slice_plane_z + m_scale;   // 2355-2358
	if (m_render_mode == ERenderMode::Slice) plane_z = -plane_z;
	const int render_mode = m_visualized_dimension > -1 ? 8 /* EncodingVis */ : (int)m_render_mode;   // 2360
	prepare_nerf_masks();
	NgpRenderExtras ex{};
	ex.render_masks = m_n_render_masks ? m_render_masks_gpu.as<NgpMask3D>() : nullptr; ex.n_render_masks = m_n_render_masks;
	ex.glow_mode = m_nerf.glow_mode; ex.glow_y_cutoff = m_nerf.glow_y_cutoff;
	if (m_envmap.resolution[0] > 0 && m_envmap.resolution[1] > 0) { ex.envmap = m_envmap.params_inference(); ex.envmap_res[0] = m_envmap.resolution[0]; ex.envmap_res[1] = m_envmap.resolution[1]; }   // 2399-2400
	if (m_nerf.render_with_lens_distortion && m_distortion.resolution[0] > 0) {   // render_grid_distortion (2370, 2401-2402)
		ex.distortion = m_distortion.params_inference(); ex.distortion_res[0] = m_distortion.resolution[0]; ex.distortion_res[1] = m_distortion.resolution[1];
	}
	ex.quilting_dims[0] = m_quilting_dims[0]; ex.quilting_dims[1] = m_quilting_dims[1];
	ex.render_mode = render_mode; ex.frame_buffer = rb.frame_buffer.as<float>();
	ex.row_begin = row_begin; ex.row_end = row_end;
	ex.tile_order = (m_nerf.render_tile_order && !(rb.res[0] & 7) && !((row_end - row_begin) & 7)) ? 1 : 0;
	float parallax_shift[3] = {m_parallax_shift[0], m_parallax_shift[1], m_parallax_shift[2]};
	if ((m_quilting_dims[0] != 1 || m_quilting_dims[1] != 1) && !(m_quilting_dims[0] == 2 && m_quilting_dims[1] == 1)) parallax_shift[2] = 1.0f / m_scale;   // testbed.cu:2703-2706 (lenticular display)
	check(ngp_hip_init_rays(m_stream, sample_index, m_tr_payload[0].as<NgpPayload>(), rb.res, focal_length, cam0.m, cam1.m, rolling_shutter, screen_center, parallax_shift, m_snap_to_pixel_centers,
	                           &m_render_aabb, m_render_aabb_to_local, m_render_near_distance, lens_mode, m_nerf.render_lens_proxy.lens_params, rb.depth_buffer.as<float>(),
	                           plane_z, m_aperture_size, m_render_camera_models.model ? &m_render_camera_models : nullptr, &ex), "init_rays");
	const NgpNetDesc* desc = m_desc_gpu.as<NgpNetDesc>();
	// the latent code / light direction presented at inference time: one row for every sample of the frame (get_inference_extra_dims, 2362)
	NgpNetVariant render_variant_storage;
	const NgpNetVariant* render_variant = net_variant(render_variant_storage, get_inference_extra_dims(), nullptr);
	// (render_mode 8 = m_visualized_dimension > -1, whatever m_render_mode says: that covers the Slice mode's activation read-out below as well)
	if (render_variant && m_netx_scalar_kernels && render_mode == (int)ERenderMode::Normals)
		throw std::runtime_error{"the Normals render mode needs the network's input gradient, which the scalar checker kernels (netx_scalar_kernels) do not have"};
	if (m_render_mode == ERenderMode::Slice) {   // 2445-2476: the network where every ray meets the slice plane; all rays of the frame are shaded
		const uint32_t n_hit = n_pixels, n_elements = (uint32_t)next_multiple(n_hit, BATCH_SIZE_GRANULARITY);
		m_tr_vis_rgba.enlarge((size_t)n_elements * 16);
		check(ngp_hip_generate_inputs_at_current_position(m_stream, n_hit, &m_aabb, m_tr_payload[0].as<NgpPayload>(), m_tr_net_in.as<NgpCoord>()), "generate_inputs_at_current_position");
		if (m_visualized_dimension == -1) {
			check(ngp_hip_nerf_inference(m_stream, desc, m_params.as<uint16_t>() /* m_network->inference: the training weights (2459) */, m_tr_net_in.as<float>(), 7, n_hit, m_tr_net_out.as<uint16_t>(), OUT_STRIDE, render_variant), "nerf_inference (slice)");
			check(ngp_hip_compute_nerf_rgba(m_stream, n_hit, m_tr_net_out.as<uint16_t>(), OUT_STRIDE, m_tr_vis_rgba.as<float>(), (int)m_nerf.rgb_activation, (int)m_nerf.density_activation, 0.01f, 0), "compute_nerf_rgba");
		} else {
			check(ngp_hip_nerf_visualize_activation(m_stream, desc, m_params.as<uint16_t>(), m_visualized_layer, (uint32_t)m_visualized_dimension, m_tr_net_in.as<float>(), 7, n_hit, m_tr_vis_rgba.as<float>(), 4, render_variant), "visualize_activation (slice)");
		}
		m_render_samples_evaluated += n_elements;
		check(ngp_hip_shade(m_stream, n_hit, m_tr_vis_rgba.as<float>(), nullptr, m_tr_payload[0].as<NgpPayload>(), m_nerf.training.linear_colors, rb.frame_buffer.as<float>(), rb.depth_buffer.as<float>(),
		                         (int)m_render_mode), "shade (slice)");
		return;
	}
	HIP_CHECK_THROW(hipMemsetAsync(m_tr_rgba[0].data(), 0, (size_t)n_pixels * 16, (hipStream_t)m_stream));
	HIP_CHECK_THROW(hipMemsetAsync(m_tr_depth[0].data(), 0, (size_t)n_pixels * 4, (hipStream_t)m_stream));
	const uint32_t min_mip = m_nerf.show_accel >= 0 ? (uint32_t)m_nerf.show_accel : 0;
	HIP_CHECK_THROW(hipMemsetAsync(hit_counter, 0, 4, (hipStream_t)m_stream));
	const bool fused_compaction = m_nerf.render_fused_compaction && std::max(1u, std::min(m_nerf.render_n_streams, 8u)) == 1;
	if (!m_nerf.brick_summary_valid && m_nerf.density_grid_bitfield.bytes() >= (size_t)GRID_CELLS / 8) {   // a bitfield that did not come from update_density_grid_mean_and_bitfield (snapshot, reset)
		m_nerf.bitfield_brick_summary.enlarge(GRID_CELLS / 64 / 32 * 4);
		check(ngp_hip_bitfield_brick_summary(m_stream, m_nerf.density_grid_bitfield.as<uint8_t>(), m_nerf.bitfield_brick_summary.as<uint32_t>()), "bitfield_brick_summary");
		m_nerf.brick_summary_valid = true;
	}
	const uint32_t* brick_summary = m_nerf.brick_summary_valid ? m_nerf.bitfield_brick_summary.as<uint32_t>() : nullptr;
	if (fused_compaction) {
		// The tracer with the compaction folded into advance_pos / composite (NgpCompactOut): per pass  march n_steps -> network -> composite+compact -> read n_alive back.
		// Same per-ray sample sequence, same pixels as the loop below (tests/test_dp_gpu.py); one launch and one 60-byte read + write per ray and pass less.
		if (!m_render_host_words) HIP_CHECK_THROW(hipHostMalloc(&m_render_host_words, 8 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
		if (m_tr_enc_ws.empty()) m_tr_enc_ws.emplace_back();
		uint32_t* alive_counter = m_tr_counters.as<uint32_t>() + 2;
		volatile uint32_t* host_alive = (volatile uint32_t*)m_render_host_words;
		hipStream_t st = (hipStream_t)m_stream;
		const bool trace = m_render_trace;   // pyngp: render_trace — the pass structure on stderr
		const auto trace_t0 = std::chrono::steady_clock::now();
		// The alive count of a pass comes back through a mailbox in host memory that the pass's last workgroup writes (NgpCompactOut::host_mailbox) and this thread polls:
		// no copy command, no stream synchronisation (an interrupt and a wake-up) between two passes.
		uint32_t* blocks_done = m_tr_counters.as<uint32_t>() + 10;
		uint32_t* mailbox = (uint32_t*)m_render_host_words + 4;
		uint32_t* mailbox_dev = nullptr;
		HIP_CHECK_THROW(hipHostGetDevicePointer((void**)&mailbox_dev, mailbox, 0));
		HIP_CHECK_THROW(hipMemsetAsync(blocks_done, 0, 4, st));
		auto compact_into = [&](int dst) {
			NgpCompactOut co;
			co.dst_rgba = m_tr_rgba[dst].as<float>(); co.dst_depth = m_tr_depth[dst].as<float>(); co.dst_payloads = m_tr_payload[dst].as<NgpPayload>();
			co.dst_final_rgba = m_tr_hit_rgba.as<float>(); co.dst_final_depth = m_tr_hit_depth.as<float>(); co.dst_final_payloads = m_tr_hit_payload.as<NgpPayload>();
			co.counter = alive_counter; co.final_counter = hit_counter;
			co.blocks_done = blocks_done; co.host_mailbox = mailbox_dev; co.sequence = ++m_render_sequence;
			return co;
		};
		auto read_alive = [&]() {
			const uint32_t want = m_render_sequence;
			const auto t0 = std::chrono::steady_clock::now();
			uint64_t word;
			for (uint32_t spins = 1; (uint32_t)((word = __atomic_load_n((const uint64_t*)mailbox, __ATOMIC_ACQUIRE)) >> 32) != want; ++spins) {
				if ((spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {   // a kernel that died posts nothing: ask the stream
					HIP_CHECK_THROW(hipStreamSynchronize(st));
					if ((uint32_t)((word = __atomic_load_n((const uint64_t*)mailbox, __ATOMIC_ACQUIRE)) >> 32) != want) throw std::runtime_error{"render: the tracer pass finished without posting its alive count"};
					break;
				}
				__builtin_ia32_pause();
			}
			return (uint32_t)word;
		};
		(void)host_alive;
		// (Cost mode paints the steps a ray took, counted per pass: every ray must take the pass's n_steps there)
		const uint32_t skip_allowance = render_mode == (int)ERenderMode::Cost ? 0u : m_nerf.render_max_skips_per_pass;
		// (n_alive * n_steps never exceeds a pass's sample budget: the workspace is sized once, not grown — a hipFree + hipMalloc — when a later pass or view needs more)
		const uint32_t pass_samples = (uint32_t)((float)n_pixels * std::min(std::max(m_nerf.render_pass_samples_factor, 1.0f), 4.0f));
		m_tr_enc_ws[0].enlarge(ngp_hip_nerf_encode_workspace_bytes(next_multiple(pass_samples, BATCH_SIZE_GRANULARITY) + BATCH_SIZE_GRANULARITY));
		HIP_CHECK_THROW(hipMemsetAsync(alive_counter, 0, 4, st));
		NgpCompactOut co = compact_into(1);
		check(ngp_hip_advance_pos(st, n_pixels, &m_render_aabb, m_render_aabb_to_local, sample_index, m_tr_payload[0].as<NgpPayload>(), m_nerf.density_grid_bitfield.as<uint8_t>(),
		                          min_mip, m_nerf.cone_angle_constant, &co, brick_summary), "advance_pos (+ compaction)");
		int cur = 1;
		uint32_t n_alive = read_alive(), i = 1;
		while (n_alive > 0 && i < MARCH_ITER) {
			const uint32_t n_steps = std::min(std::max(pass_samples / n_alive, 1u), m_nerf.render_steps_cap());   // NerfTracer::trace (2231), cap raised (see below)
			if (trace) fprintf(stderr, "render pass i=%u n_alive=%u n_steps=%u t=%.3f ms\n", i, n_alive, n_steps, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - trace_t0).count());
			NgpPayload* payloads = m_tr_payload[cur].as<NgpPayload>();
			NgpCoord* net_in = m_tr_net_in.as<NgpCoord>();
			uint16_t* net_out = m_tr_net_out.as<uint16_t>();
			check(ngp_hip_generate_next_inputs(st, n_alive, &m_render_aabb, &m_aabb, payloads, net_in, n_steps, m_nerf.density_grid_bitfield.as<uint8_t>(), min_mip, m_nerf.cone_angle_constant,
			                                   skip_allowance, brick_summary, alive_counter /* read back above; composite below bumps it */), "generate_next_inputs");
			const uint32_t n_elements = next_multiple(n_alive * n_steps, BATCH_SIZE_GRANULARITY);
			m_tr_enc_ws[0].enlarge(ngp_hip_nerf_encode_workspace_bytes(std::max(n_elements, n_pixels)));
			if (m_nerf.render_fused_network) check(ngp_hip_nerf_inference(st, desc, m_inference_params.as<uint16_t>(), (const float*)net_in, 7, n_elements, net_out, OUT_STRIDE, render_variant), "nerf_inference (render, fused)");
			else check(ngp_hip_nerf_inference_ws(st, desc, m_inference_params.as<uint16_t>(), (const float*)net_in, 7, n_elements, net_out, OUT_STRIDE, m_tr_enc_ws[0].data(), m_tr_enc_ws[0].bytes(), render_variant), "nerf_inference (render)");
			m_render_samples_evaluated += n_elements;
			if (render_mode == (int)ERenderMode::Normals) {
				const uint32_t n_grad = (uint32_t)next_multiple(n_elements, 256u);
				m_tr_vis_scratch.enlarge(ngp_hip_nerf_input_gradient_scratch_bytes(n_grad));
				check(ngp_hip_nerf_input_gradient(st, desc, &m_desc, m_inference_params.as<uint16_t>(), 3, (float*)net_in, 7, n_grad, m_tr_vis_scratch.data(), m_tr_vis_scratch.bytes(), render_variant), "nerf_input_gradient (normals)");
			} else if (render_mode == 8) {
				check(ngp_hip_nerf_visualize_activation(st, desc, m_inference_params.as<uint16_t>(), m_visualized_layer, (uint32_t)m_visualized_dimension, (const float*)net_in, 7, n_elements, (float*)net_in, 7, render_variant), "visualize_activation");
			}
			co = compact_into(cur ^ 1);
			check(ngp_hip_composite(st, n_alive, i, &m_aabb, cam1.m, m_tr_rgba[cur].as<float>(), m_tr_depth[cur].as<float>(), payloads, net_in, net_out, OUT_STRIDE, n_steps,
			                        (int)m_nerf.rgb_activation, (int)m_nerf.density_activation, m_nerf.render_min_transmittance, render_mode,
			                        1.0f / m_nerf.training.dataset.scale /* 2415 */, m_nerf.show_accel, &ex, &co), "composite (+ compaction)");
			i += n_steps;
			cur ^= 1;
			n_alive = read_alive();
		}
		uint32_t n_hit = 0;
		HIP_CHECK_THROW(hipMemcpyAsync(&n_hit, hit_counter, 4, hipMemcpyDeviceToHost, st));
		sync();
		check(ngp_hip_shade(m_stream, n_hit, m_tr_hit_rgba.as<float>(), m_tr_hit_depth.as<float>(), m_tr_hit_payload.as<NgpPayload>(), m_nerf.training.linear_colors,
		                    rb.frame_buffer.as<float>(), rb.depth_buffer.as<float>(), (int)m_render_mode), "shade");
		return;
	}
	check(ngp_hip_advance_pos(m_stream, n_pixels, &m_render_aabb, m_render_aabb_to_local, sample_index, m_tr_payload[0].as<NgpPayload>(), m_nerf.density_grid_bitfield.as<uint8_t>(),
	                          min_mip, m_nerf.cone_angle_constant, nullptr, brick_summary), "advance_pos");

	// NerfTracer::trace (2140-2267).  The reference walks all rays of the frame in lock step on one stream: compact -> read n_alive back ->
	// march n_steps -> network -> composite, ~40 times per frame, and every pass pays a host round trip plus a march that is bound by the
	// latency of its longest ray.  Rays are independent, so the frame CAN be cut into K contiguous pixel ranges, each with its own stream and
	// counters (while the host waits for one range's compaction result, the other ranges' passes run; same per-ray arithmetic, same image).
	// Measured on MI355X / ROCm 7.0 the cross-queue scheduling costs more than the overlap buys (K = 1: 16.1 ms, 2: 27.5 ms, 3: 31.9 ms
	// per 800x800 frame), so K defaults to 1 and the knob stays for re-measurement.
	const uint32_t K = std::max(1u, std::min(m_nerf.render_n_streams, 8u));
	while (m_render_streams.size() < K) { hipStream_t st; HIP_CHECK_THROW(hipStreamCreate(&st)); m_render_streams.push_back(st); }
	if (!m_render_event) { hipEvent_t e; HIP_CHECK_THROW(hipEventCreateWithFlags(&e, hipEventDisableTiming)); m_render_event = e; }
	if (!m_render_host_words) HIP_CHECK_THROW(hipHostMalloc(&m_render_host_words, 8 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
	m_tr_counters.enlarge((2 + 8) * 4);
	while (m_tr_enc_ws.size() < K) m_tr_enc_ws.emplace_back();
	struct Part { uint32_t start, count, n_alive, i, dbi; bool done; };
	std::vector<Part> parts(K);
	const uint32_t per_part = next_multiple((n_pixels + K - 1) / K, BATCH_SIZE_GRANULARITY);
	for (uint32_t p = 0; p < K; ++p) {
		const uint32_t start = std::min(p * per_part, n_pixels);
		parts[p] = Part{start, std::min(per_part, n_pixels - start), 0, 1, 0, false};
		parts[p].n_alive = parts[p].count;
		parts[p].done = parts[p].count == 0;
	}
	HIP_CHECK_THROW(hipEventRecord((hipEvent_t)m_render_event, (hipStream_t)m_stream));
	for (uint32_t p = 0; p < K; ++p) HIP_CHECK_THROW(hipStreamWaitEvent((hipStream_t)m_render_streams[p], (hipEvent_t)m_render_event, 0));
	uint32_t* alive_counters = m_tr_counters.as<uint32_t>() + 2;
	volatile uint32_t* host_alive = (volatile uint32_t*)m_render_host_words;
	const bool trace = m_render_trace;
	auto buf = [&](DeviceBuffer& b, size_t elem_bytes, uint32_t start) { return (char*)b.data() + (size_t)start * elem_bytes; };
	bool any = true;
	while (any) {
		// stage A: queue the compaction of every active range and the read-back of its alive count
		for (uint32_t p = 0; p < K; ++p) {
			Part& pt = parts[p];
			if (pt.done) continue;
			hipStream_t st = (hipStream_t)m_render_streams[p];
			const int cur = (pt.dbi + 1) % 2, tmp = pt.dbi % 2;
			HIP_CHECK_THROW(hipMemsetAsync(alive_counters + p, 0, 4, st));
			check(ngp_hip_compact_rays(st, pt.n_alive, (float*)buf(m_tr_rgba[tmp], 16, pt.start), (float*)buf(m_tr_depth[tmp], 4, pt.start), (NgpPayload*)buf(m_tr_payload[tmp], sizeof(NgpPayload), pt.start),
			                           (float*)buf(m_tr_rgba[cur], 16, pt.start), (float*)buf(m_tr_depth[cur], 4, pt.start), (NgpPayload*)buf(m_tr_payload[cur], sizeof(NgpPayload), pt.start),
			                           m_tr_hit_rgba.as<float>(), m_tr_hit_depth.as<float>(), m_tr_hit_payload.as<NgpPayload>(), alive_counters + p, hit_counter), "compact_rays");
			HIP_CHECK_THROW(hipMemcpyAsync((void*)(host_alive + p), alive_counters + p, 4, hipMemcpyDeviceToHost, st));
		}
		// stage B: per range, wait for its count only, then queue its march / network / composite
		any = false;
		for (uint32_t p = 0; p < K; ++p) {
			Part& pt = parts[p];
			if (pt.done) continue;
			hipStream_t st = (hipStream_t)m_render_streams[p];
			HIP_CHECK_THROW(hipStreamSynchronize(st));
			const int cur = (pt.dbi + 1) % 2;
			++pt.dbi;
			pt.n_alive = host_alive[p];
			if (pt.n_alive == 0 || pt.i >= MARCH_ITER) { pt.done = true; continue; }
			any = true;
			// NerfTracer::trace (2231): clamp(n_rays_initialized / n_alive, 1, 8).  The per-ray sample sequence does not depend on how it is cut
			// into passes and n_alive * n_steps <= n_rays keeps every buffer as sized for 8, so the cap is raised once few rays are left.
			const uint32_t n_steps = std::min(std::max(pt.count / pt.n_alive, 1u), m_nerf.render_steps_cap());
			if (trace) fprintf(stderr, "render pass part=%u i=%u n_alive=%u n_steps=%u\n", p, pt.i, pt.n_alive, n_steps);
			NgpPayload* payloads = (NgpPayload*)buf(m_tr_payload[cur], sizeof(NgpPayload), pt.start);
			NgpCoord* net_in = (NgpCoord*)buf(m_tr_net_in, 8 * sizeof(NgpCoord), pt.start);
			uint16_t* net_out = (uint16_t*)buf(m_tr_net_out, 8 * OUT_STRIDE * 2, pt.start);
			check(ngp_hip_generate_next_inputs(st, pt.n_alive, &m_render_aabb, &m_aabb, payloads, net_in, n_steps, m_nerf.density_grid_bitfield.as<uint8_t>(), min_mip, m_nerf.cone_angle_constant, 0, brick_summary, nullptr), "generate_next_inputs");
			const uint32_t n_elements = next_multiple(pt.n_alive * n_steps, BATCH_SIZE_GRANULARITY);
			// inference on the EMA weights (use_inference_params defaults to true at testbed_nerf.cu:2223)
			m_tr_enc_ws[p].enlarge(ngp_hip_nerf_encode_workspace_bytes(std::max(n_elements, pt.count)));
			if (m_nerf.render_fused_network) check(ngp_hip_nerf_inference(st, desc, m_inference_params.as<uint16_t>(), (const float*)net_in, 7, n_elements, net_out, OUT_STRIDE, render_variant), "nerf_inference (render, fused)");
			else check(ngp_hip_nerf_inference_ws(st, desc, m_inference_params.as<uint16_t>(), (const float*)net_in, 7, n_elements, net_out, OUT_STRIDE, m_tr_enc_ws[p].data(), m_tr_enc_ws[p].bytes(), render_variant), "nerf_inference (render)");
			m_render_samples_evaluated += n_elements;
			if (render_mode == (int)ERenderMode::Normals) {   // 2225-2226: network.input_gradient(stream, 3, positions, positions) — on the inference weights like the pass above
				const uint32_t n_grad = (uint32_t)next_multiple(n_elements, 256u);
				m_tr_vis_scratch.enlarge(ngp_hip_nerf_input_gradient_scratch_bytes(n_grad));
				check(ngp_hip_nerf_input_gradient(st, desc, &m_desc, m_inference_params.as<uint16_t>(), 3, (float*)net_in, 7, n_grad, m_tr_vis_scratch.data(), m_tr_vis_scratch.bytes(), render_variant), "nerf_input_gradient (normals)");
			} else if (render_mode == 8) {                    // 2227-2228: network.visualize_activation(stream, layer, dim, positions, positions)
				check(ngp_hip_nerf_visualize_activation(st, desc, m_inference_params.as<uint16_t>(), m_visualized_layer, (uint32_t)m_visualized_dimension, (const float*)net_in, 7, n_elements, (float*)net_in, 7, render_variant), "visualize_activation");
			}
			check(ngp_hip_composite(st, pt.n_alive, pt.i, &m_aabb, cam1.m, (float*)buf(m_tr_rgba[cur], 16, pt.start), (float*)buf(m_tr_depth[cur], 4, pt.start), payloads, net_in, net_out, OUT_STRIDE, n_steps,
			                           (int)m_nerf.rgb_activation, (int)m_nerf.density_activation, m_nerf.render_min_transmittance, render_mode,
			                           1.0f / m_nerf.training.dataset.scale /* 2415 */, m_nerf.show_accel, &ex, nullptr), "composite");
			pt.i += n_steps;
		}
	}
	uint32_t n_hit = 0;
	HIP_CHECK_THROW(hipMemcpyAsync(&n_hit, hit_counter, 4, hipMemcpyDeviceToHost, (hipStream_t)m_stream));
	sync();
	check(ngp_hip_shade(m_stream, n_hit, m_tr_hit_rgba.as<float>(), m_tr_hit_depth.as<float>(), m_tr_hit_payload.as<NgpPayload>(), m_nerf.training.linear_colors,
	                         rb.frame_buffer.as<float>(), rb.depth_buffer.as<float>(), (int)m_render_mode), "shade");
}

// ---- Blender multi-NeRF requests -------------------------------------------------------------------------------------------
void Testbed::bl_render_frame(RenderBuffer& rb, const RenderRequest& request) {  // testbed.cu:2675-2693
	if (!m_renderer) m_renderer.reset(new NerfRenderer());
	rb.frame_buffer.memset(0, m_stream);   // CudaRenderBuffer::clear_frame
	rb.depth_buffer.memset(0, m_stream);
	m_renderer->trace = m_render_trace;
	m_renderer->fused_passes = m_bl_fused_passes; m_renderer->reference_schedule = m_bl_reference_schedule; m_renderer->max_skips_per_pass = m_bl_max_skips_per_pass;
	m_renderer->pass_samples_factor = m_bl_pass_samples_factor; m_renderer->max_steps_per_pass = m_bl_max_steps_per_pass;
	m_bl_render_samples = m_renderer->render(rb, request, m_stream);
	m_bl_render_passes = m_renderer->last_n_passes;
	rb.color_space = request.output.color_space;
	rb.tonemap_curve = request.output.tonemap_curve;
	if (rb.spp == 0) rb.accumulate_buffer.memset(0, m_stream);
	check(ngp_hip_accumulate(m_stream, rb.res, rb.frame_buffer.as<float>(), rb.accumulate_buffer.as<float>(), (float)rb.spp, (int)rb.color_space), "accumulate");
	++rb.spp;
	check(ngp_hip_tonemap(m_stream, rb.res, request.output.exposure, request.output.background_color, rb.accumulate_buffer.as<float>(), (int)rb.color_space,
	                      (int)request.output.color_space, (int)rb.tonemap_curve, 0, rb.surface.as<float>()), "tonemap");
	sync();
}

void Testbed::bl_start_async(std::function<void()> job) {
	{ std::lock_guard<std::mutex> lock(m_render_mutex); ++m_render_workers; }
	try {
		std::thread([this, job]() {
			try { (void)hipSetDevice(m_device); job(); } catch (...) {}   // HIP's current device is per thread
			{ std::lock_guard<std::mutex> lock(m_render_mutex); --m_render_workers; }
			m_render_cv.notify_all();
		}).detach();
	} catch (...) {
		{ std::lock_guard<std::mutex> lock(m_render_mutex); --m_render_workers; }
		throw;
	}
}
void Testbed::bl_wait_for_renders() {
	std::unique_lock<std::mutex> lock(m_render_mutex);
	m_render_cv.wait(lock, [this]() { return m_render_workers == 0; });
}

bool Testbed::bl_try_begin_render() { bool expected = false; return m_currently_rendering.compare_exchange_strong(expected, true); }
void Testbed::bl_end_render() { m_currently_rendering.store(false); }

bool Testbed::bl_request_nerf_render_sync(const RenderRequest& request, float* out) {  // python_api.cu:233-260; out: H * W * 4 floats, untouched (false) while another render is in flight (:235-237)
	const int w = request.output.resolution[0], h = request.output.resolution[1];
	if (!bl_try_begin_render()) return false;
	try {
		if (m_autofocus) autofocus();   // python_api.cu:240
		RenderBuffer& rb = m_bl_render_surface;
		rb.resize(w, h);
		rb.reset_accumulation();
		bl_render_frame(rb, request);
		download(rb.surface.data(), (size_t)w * h * 16, out);
	} catch (...) { bl_end_render(); throw; }
	bl_end_render();
	return true;
}
std::vector<float> Testbed::bl_request_nerf_render_sync(const RenderRequest& request) {
	std::vector<float> out((size_t)request.output.resolution[0] * request.output.resolution[1] * 4, 0.f);
	bl_request_nerf_render_sync(request, out.data());
	return out;
}

// ---- snapshots (testbed.cu:3006-3106; schema in snapshot.h) ---------------------------------------------------------------
static Json buffer_to_json_binary(const DeviceBuffer& b, size_t bytes) {
	std::vector<uint8_t> host(bytes);
	if (bytes) b.copy_to_host(host.data(), bytes);
	return Json::binary(host.data(), host.size());
}

void Testbed::save_snapshot(const std::string& path, bool include_optimizer_state) {
	if (m_n_params == 0) throw std::runtime_error{"save_snapshot: no network"};
	drop_prefetch();
	sync();
	require_inference_params("save_snapshot", false);   // the snapshot's params_binary are the inference (Ema) weights: a rank saves alone, so no collective here
	if (include_optimizer_state && m_dp_state_stale)   // sharded data-parallel steps left master weights / moments current only inside this rank's shard
		throw std::runtime_error{"save_snapshot(include_optimizer_state=True): the fp32 optimizer state is sharded over the data-parallel ranks — call dp_gather_optimizer_state() on ALL ranks first (a collective), then save on whichever rank"};
	Json snapshot = Json::object();
	// Trainer::serialize [tcnn]: the inference (EMA) weights in the network precision
	snapshot["n_params"] = Json((unsigned long long)m_n_params);
	snapshot["params_type"] = Json("__half");
	snapshot["params_binary"] = buffer_to_json_binary(m_inference_params, m_n_params * 2);
	if (include_optimizer_state) {
		// Ema o ExponentialDecay o Adam, nested like the config (configs/nerf/base.json:5-22)
		Json adam = Json::object();
		adam["otype"] = Json("Adam");
		adam["current_step"] = Json(m_optimizer_step);
		adam["base_learning_rate"] = Json((double)m_base_learning_rate);
		adam["first_moments_binary"] = buffer_to_json_binary(m_first_moments, m_n_params * 4);
		adam["second_moments_binary"] = buffer_to_json_binary(m_second_moments, m_n_params * 4);
		Json decay = Json::object();
		decay["otype"] = Json("ExponentialDecay");
		decay["learning_rate"] = Json((double)m_learning_rate);
		decay["nested"] = adam;
		Json ema = Json::object();
		ema["otype"] = Json("Ema");
		ema["ema_step"] = Json(m_optimizer_step);
		ema["ema_binary"] = buffer_to_json_binary(m_ema, m_n_params * 4);
		ema["nested"] = decay;
		// full-precision master weights and the training (non-EMA) fp16 weights, so that a resumed run continues exactly
		ema["params_full_precision_binary"] = buffer_to_json_binary(m_master, m_n_params * 4);
		snapshot["optimizer"] = ema;
	}
	snapshot["version"] = Json(1);
	snapshot["density_grid_size"] = Json((unsigned)NGP_NERF_GRIDSIZE);
	{
		const size_t n_cells = m_nerf.density_grid.bytes() / 4;
		std::vector<float> grid(n_cells);
		if (n_cells) m_nerf.density_grid.copy_to_host(grid.data(), n_cells * 4);
		std::vector<uint16_t> grid_fp16(n_cells);
		for (size_t i = 0; i < n_cells; ++i) grid_fp16[i] = float_to_half_bits(grid[i]);
		snapshot["density_grid_binary"] = Json::binary(grid_fp16.data(), n_cells * 2);
	}
	Json nerf = Json::object();
	nerf["aabb_scale"] = Json(m_nerf.training.dataset.aabb_scale);
	Json rgb = Json::object();
	const NerfCounters& c = m_nerf.training.counters_rgb;
	rgb["rays_per_batch"] = Json(c.rays_per_batch);
	rgb["measured_batch_size"] = Json(c.measured_batch_size);
	rgb["measured_batch_size_before_compaction"] = Json(c.measured_batch_size_before_compaction);
	nerf["rgb"] = rgb;
	nerf["dataset"] = dataset_to_json(m_nerf.training.dataset);
	snapshot["nerf"] = nerf;
	snapshot["training_step"] = Json(m_training_step);
	snapshot["loss"] = Json((double)m_loss_scalar);
	snapshot["aabb"] = aabb_to_json(m_aabb);
	snapshot["bounding_radius"] = Json((double)m_bounding_radius);

	m_network_config["snapshot"] = snapshot;
	m_network_config_path = path;
	m_network_config.to_msgpack_file(path);
}

void Testbed::load_snapshot(const std::string& path) {
	join_side_ema();
	Json config = load_network_config(path);
	if (!config.contains("snapshot")) throw std::runtime_error{"File " + path + " does not contain a snapshot."};
	const Json& snapshot = config["snapshot"];
	if (snapshot.value("version", 0) < 1) throw std::runtime_error{"Snapshot uses an old format."};
	if (snapshot.contains("aabb")) m_aabb = aabb_from_json(snapshot["aabb"]);
	m_bounding_radius = (float)snapshot.value("bounding_radius", (double)m_bounding_radius);
	if ((uint32_t)snapshot.at("density_grid_size").number() != NGP_NERF_GRIDSIZE) throw std::runtime_error{"Incompatible grid size."};

	NerfTraining& tr = m_nerf.training;
	NerfCounters& c = tr.counters_rgb;
	const Json& rgb = snapshot.at("nerf").at("rgb");
	const uint32_t rays_per_batch = (uint32_t)rgb.at("rays_per_batch").number();
	const uint32_t measured = (uint32_t)rgb.at("measured_batch_size").number();
	const uint32_t measured_before = (uint32_t)rgb.at("measured_batch_size_before_compaction").number();

	// without a dataset of our own, take the metadata from the snapshot and render from that alone (3071-3078)
	if (!m_training_data_available && snapshot["nerf"].contains("dataset")) {
		dataset_from_json(snapshot["nerf"]["dataset"], tr.dataset);
	} else if (snapshot["nerf"].contains("aabb_scale")) {
		tr.dataset.aabb_scale = (int)snapshot["nerf"]["aabb_scale"].number();
	}
	load_nerf_post();
	if (!m_training_data_available) { tr.n_images_for_training = 0; m_train = false; }  // metadata only: no pixels to train on

	std::vector<float> grid;
	snapshot_read_density_grid(snapshot, grid);
	const size_t expected = (size_t)GRID_CELLS * (m_nerf.max_cascade + 1);
	if (grid.size() != expected && !grid.empty()) throw std::runtime_error{"Incompatible number of grid cascades."};

	m_network_config_path = path;
	m_network_config = config;
	reset_network(false);
	c.rays_per_batch = rays_per_batch; c.measured_batch_size = measured; c.measured_batch_size_before_compaction = measured_before;

	m_nerf.density_grid.resize(expected * 4);
	m_nerf.density_grid_bitfield.resize((size_t)GRID_CELLS);
	m_nerf.density_grid_mean.resize(4);
	if (!grid.empty()) {
		m_nerf.density_grid.copy_from_host(grid.data(), grid.size() * 4);
		update_density_grid_mean_and_bitfield();
	} else {
		m_nerf.density_grid.memset(0, m_stream); m_nerf.density_grid_bitfield.memset(0, m_stream); m_nerf.density_grid_mean.memset(0, m_stream);
		m_nerf.brick_summary_valid = false;
	}

	m_training_step = (uint32_t)snapshot.at("training_step").number();
	m_loss_scalar = (float)snapshot.at("loss").number();

	// Trainer::deserialize [tcnn]
	std::vector<uint16_t> p16; std::vector<float> p32;
	snapshot_read_params(snapshot, p16, p32);
	if (p16.size() != m_n_params) throw std::runtime_error{"Snapshot has " + std::to_string(p16.size()) + " parameters, the network config needs " + std::to_string(m_n_params) + "."};
	m_params.copy_from_host(p16.data(), m_n_params * 2);
	m_inference_params.copy_from_host(p16.data(), m_n_params * 2);
	m_master.copy_from_host(p32.data(), m_n_params * 4);
	if (snapshot.contains("optimizer")) {
		// tiny-cuda-nn's Trainer::serialize is not in the reference tree: the key names below are this build's own layout (host/snapshot.h).  A block
		// written by the CUDA build that names its buffers differently restores nothing — say so instead of resuming silently on zero moments.
		const Json* o = &snapshot["optimizer"];
		uint32_t restored = 0, blocks = 0;
		auto take = [&](const Json& j, const char* key, DeviceBuffer& dst) {
			if (!j.contains(key)) return false;
			if (j.at(key).bin().size() != m_n_params * 4) { fprintf(stderr, "load_snapshot: optimizer.%s has %zu bytes, expected %zu: ignored\n", key, j.at(key).bin().size(), m_n_params * 4); return false; }
			dst.copy_from_host(j.at(key).bin().data(), m_n_params * 4);
			++restored;
			return true;
		};
		while (true) {
			const std::string otype = o->value("otype", "");
			++blocks;
			if (otype == "Ema") {
				take(*o, "ema_binary", m_ema);
				if (take(*o, "params_full_precision_binary", m_master)) {
					const std::vector<uint8_t>& mb = o->at("params_full_precision_binary").bin();
					std::vector<uint16_t> train16(m_n_params);
					const float* mf = (const float*)mb.data();
					for (size_t i = 0; i < m_n_params; ++i) train16[i] = float_to_half_bits(mf[i]);
					m_params.copy_from_host(train16.data(), m_n_params * 2);
				}
			} else if (otype == "ExponentialDecay") {
				m_learning_rate = (float)o->value("learning_rate", (double)m_learning_rate);
			} else if (otype == "Adam") {
				m_optimizer_step = (uint32_t)o->value("current_step", 0);
				take(*o, "first_moments_binary", m_first_moments);
				take(*o, "second_moments_binary", m_second_moments);
			} else {
				fprintf(stderr, "load_snapshot: optimizer block of type '%s' is not understood: skipped\n", otype.c_str());
			}
			if (!o->contains("nested")) break;
			o = &o->at("nested");
		}
		if (restored == 0) fprintf(stderr, "load_snapshot: the snapshot carries an optimizer state (%u nested blocks) but none of its buffers could be restored: training resumes with fresh Adam moments\n", blocks);
	}
	sync();
}

} // namespace ngp
