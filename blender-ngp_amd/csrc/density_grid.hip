// density_grid.hip — occupancy ("density") grid maintenance for gfx950.
// Replaces src/testbed_nerf.cu:369-416 (mark_untrained_density_grid), 465-494 (generate_grid_samples_nerf_nonuniform),
// 496-512 (splat), 532-555 (ema), 563-610 (grid_to_bitfield, bitfield_max_pool), 2851-2852 (mean via reduce_sum).
// All index math is exact (built -ffp-contract=off): bit-exact against the CPU oracle.
#include "ngp_device.cuh"
#include <string.h>
#include <stdio.h>

namespace ngp {

static thread_local char g_last_error[512] = "";
void set_last_error(const char* what, hipError_t e) {
	snprintf(g_last_error, sizeof(g_last_error), "%s: %s (%d)", what, e == hipSuccess ? "ok" : hipGetErrorString(e), (int)e);
}

__global__ void mark_untrained_kernel(uint32_t n_elements, float* __restrict__ grid_out, uint32_t n_training_images,
                                      const NgpImageMeta* __restrict__ metadata, const NgpXForm* __restrict__ xforms, bool clear_visible_voxels) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const uint32_t level = i / NGP_NERF_GRID_N_CELLS;
	const uint32_t pos_idx = i % NGP_NERF_GRID_N_CELLS;
	const uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
	const float s = __uint_as_float((127u + level) << 23); // scalbnf(1, level)
	const v3 pos = mk((((float)x + 0.5f) / 128.0f - 0.5f) * s + 0.5f, (((float)y + 0.5f) / 128.0f - 0.5f) * s + 0.5f, (((float)z + 0.5f) / 128.0f - 0.5f) * s + 0.5f);
	const float voxel_radius = 0.5f * SQRT3() * s / 128.0f;
	int count = 0;
	for (uint32_t j = 0; j < n_training_images; ++j) {
		const NgpImageMeta& md = metadata[j];
		if (md.lens_mode == 2 || md.lens_mode == 3) { count++; break; }
		const float half_resx = (float)md.res[0] * 0.5f, half_resy = (float)md.res[1] * 0.5f;
		const float* xf = xforms[j].start;
		const v3 ploc = pos - col(xf, 3);
		const float px = dot(ploc, col(xf, 0)), py = dot(ploc, col(xf, 1)), pz = dot(ploc, col(xf, 2));
		if (pz > 0.f) {
			if (fabsf(px) - voxel_radius < pz / md.focal_length[0] * half_resx && fabsf(py) - voxel_radius < pz / md.focal_length[1] * half_resy) {
				count++;
				break;
			}
		}
	}
	if (clear_visible_voxels || (grid_out[i] < 0) != (count <= 0)) grid_out[i] = (count > 0) ? 0.f : -1.f;
}

__global__ void grid_samples_nonuniform_kernel(uint32_t n_elements, Pcg32 rng, uint32_t step, Aabb aabb, const float* __restrict__ grid_in,
                                               float* __restrict__ out_pos, uint32_t* __restrict__ indices, uint32_t n_cascades, float thresh) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	rng.advance((uint64_t)(i * 4u));
	const uint32_t level = (uint32_t)(rng.next_float() * (float)n_cascades) % n_cascades;
	uint32_t idx = 0;
	for (uint32_t j = 0; j < 10; ++j) {
		idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % NGP_NERF_GRID_N_CELLS;
		idx += level * NGP_NERF_GRID_N_CELLS;
		if (grid_in[idx] > thresh) break;
	}
	const uint32_t pos_idx = idx % NGP_NERF_GRID_N_CELLS;
	const uint32_t x = morton3D_invert(pos_idx >> 0), y = morton3D_invert(pos_idx >> 1), z = morton3D_invert(pos_idx >> 2);
	const float rx = rng.next_float(), ry = rng.next_float(), rz = rng.next_float();
	const float s = __uint_as_float((127u + level) << 23);
	const v3 pos = mk((((float)x + rx) / 128.0f - 0.5f) * s + 0.5f, (((float)y + ry) / 128.0f - 0.5f) * s + 0.5f, (((float)z + rz) / 128.0f - 0.5f) * s + 0.5f);
	const v3 w = aabb_relative_pos(aabb, pos);
	out_pos[3 * i + 0] = w.x; out_pos[3 * i + 1] = w.y; out_pos[3 * i + 2] = w.z;
	indices[i] = idx;
}

__global__ void splat_max_kernel(uint32_t n_elements, const uint32_t* __restrict__ indices, const uint16_t* __restrict__ network_output,
                                 float* __restrict__ grid_out, int density_activation) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const float mlp = network_to_density(h2f(network_output[i]), density_activation);
	const float optical_thickness = mlp * MIN_CONE_STEPSIZE();
	// positive floats order like their bit patterns: uint atomicMax (testbed_nerf.cu:509-511)
	atomicMax((uint32_t*)&grid_out[indices[i]], __float_as_uint(optical_thickness));
}

__global__ void ema_kernel(uint32_t n_elements, float decay, float* __restrict__ grid_out, const float* __restrict__ grid_in) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const float prev_val = grid_out[i];
	grid_out[i] = (prev_val < 0.f) ? prev_val : fmaxf(prev_val * decay, grid_in[i]);
}

// sum_i f(in[i]) with f = max(v,0)/n (MODE 0) or identity (MODE 1); wave shuffle -> LDS -> one atomicAdd per block
template <int MODE>
__global__ void __launch_bounds__(256) reduce_sum_kernel(const float* __restrict__ in, uint32_t n, float scale, float* __restrict__ out) {
	__shared__ float partial[4];
	float acc = 0.0f;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
		const float v = in[i];
		acc += MODE == 0 ? fmaxf(v, 0.f) * scale : v;
	}
	for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
	if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = acc;
	__syncthreads();
	if (threadIdx.x == 0) atomicAdd(out, partial[0] + partial[1] + partial[2] + partial[3]);
}

__global__ void grid_to_bitfield_kernel(uint32_t n_elements, uint32_t n_nonzero_elements, const float* __restrict__ grid,
                                        uint8_t* __restrict__ bitfield, const float* __restrict__ mean_density_ptr) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	if (i >= n_nonzero_elements) { bitfield[i] = 0; return; }
	const float thresh = fminf(MIN_OPTICAL_THICKNESS(), *mean_density_ptr);
	const float4 a = ((const float4*)grid)[2 * i], b = ((const float4*)grid)[2 * i + 1];
	uint8_t bits = 0;
	bits |= a.x > thresh ? 1 : 0; bits |= a.y > thresh ? 2 : 0; bits |= a.z > thresh ? 4 : 0; bits |= a.w > thresh ? 8 : 0;
	bits |= b.x > thresh ? 16 : 0; bits |= b.y > thresh ? 32 : 0; bits |= b.z > thresh ? 64 : 0; bits |= b.w > thresh ? 128 : 0;
	bitfield[i] = bits;
}

// testbed_nerf.cu:589-610.  Thread i reads 8 bytes of the finer level (64 cells = 8 coarse cells) and produces ONE byte of the coarser
// level: bit j is set if fine byte j has any cell set.  The target is the byte whose (x, y, z) byte-coordinates are those of i shifted
// by 16 (the finer cascade covers the central half of the coarser one); targets are distinct across threads, so the `|=` is race-free.
__global__ void bitfield_max_pool_kernel(uint32_t n_elements, const uint8_t* __restrict__ prev_level, uint8_t* __restrict__ next_level) {
	const uint32_t i = threadIdx.x + blockIdx.x * blockDim.x;
	if (i >= n_elements) return;
	const uint64_t eight = ((const uint64_t*)prev_level)[i];
	uint8_t bits = 0;
#pragma unroll
	for (int j = 0; j < 8; ++j) bits |= ((eight >> (8 * j)) & 0xffu) ? (uint8_t)(1u << j) : 0;
	const uint32_t x = morton3D_invert(i >> 0) + 16, y = morton3D_invert(i >> 1) + 16, z = morton3D_invert(i >> 2) + 16;
	next_level[morton3D(x, y, z)] |= bits;
}

} // namespace ngp

using namespace ngp;

extern "C" {

int ngp_hip_abi_version(void) { return NGP_HIP_ABI_VERSION; }
const char* ngp_hip_last_error(void) { return g_last_error; }

int ngp_hip_mark_untrained_density_grid(void* stream, uint32_t n_elements, float* grid_out, uint32_t n_training_images,
                                        const NgpImageMeta* metadata, const NgpXForm* xforms, int clear_visible_voxels) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(mark_untrained_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, grid_out, n_training_images, metadata, xforms, clear_visible_voxels != 0);
	NGP_LAUNCH_CHECK("mark_untrained_kernel");
	return 0;
}

int ngp_hip_generate_grid_samples_nonuniform(void* stream, uint32_t n_elements, uint64_t rng_state, uint64_t rng_inc, uint32_t step,
                                             const NgpAabb* aabb_host, const float* grid_in, float* out_pos, uint32_t* indices, uint32_t n_cascades, float thresh) {
	if (!n_elements) return 0;
	Pcg32 rng; rng.state = rng_state; rng.inc = rng_inc;
	hipLaunchKernelGGL(grid_samples_nonuniform_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, rng, step, aabb_from_host(aabb_host), grid_in, out_pos, indices, n_cascades, thresh);
	NGP_LAUNCH_CHECK("grid_samples_nonuniform_kernel");
	return 0;
}

int ngp_hip_splat_grid_samples_max(void* stream, uint32_t n_elements, const uint32_t* indices, const uint16_t* network_output, float* grid_out, int density_activation) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(splat_max_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, indices, network_output, grid_out, density_activation);
	NGP_LAUNCH_CHECK("splat_max_kernel");
	return 0;
}

int ngp_hip_ema_grid_samples(void* stream, uint32_t n_elements, float decay, float* grid_out, const float* grid_in) {
	if (!n_elements) return 0;
	hipLaunchKernelGGL(ema_kernel, dim3(div_up(n_elements, 256)), dim3(256), 0, (hipStream_t)stream, n_elements, decay, grid_out, grid_in);
	NGP_LAUNCH_CHECK("ema_kernel");
	return 0;
}

int ngp_hip_density_grid_mean(void* stream, const float* grid, uint32_t n_elements, float* mean_out) {
	NGP_HIP_TRY(hipMemsetAsync(mean_out, 0, sizeof(float), (hipStream_t)stream));
	uint32_t blocks = div_up(n_elements, 256 * 16); blocks = blocks > 1024 ? 1024 : (blocks ? blocks : 1);
	hipLaunchKernelGGL(reduce_sum_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, grid, n_elements, 1.0f / (float)n_elements, mean_out);
	NGP_LAUNCH_CHECK("reduce_sum_kernel<0>");
	return 0;
}

int ngp_hip_reduce_sum_f32(void* stream, const float* in, uint32_t n, float* out) {
	NGP_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream));
	if (!n) return 0;
	uint32_t blocks = div_up(n, 256 * 16); blocks = blocks > 1024 ? 1024 : (blocks ? blocks : 1);
	hipLaunchKernelGGL(reduce_sum_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, n, 1.0f, out);
	NGP_LAUNCH_CHECK("reduce_sum_kernel<1>");
	return 0;
}

__global__ void gather_words_kernel(const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* dst) {
	if (threadIdx.x == 0) {
		dst[0] = a ? *a : 0u; dst[1] = b ? *b : 0u; dst[2] = c ? *c : 0u; dst[3] = d ? *d : 0u;
		__threadfence_system();
	}
}

__global__ void post_words_kernel(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t tag, uint32_t* dst, uint32_t* zero_words, uint32_t n_zero_words, double* sum3_dev) {
	if (threadIdx.x == 0) {
		const uint32_t va = a ? *a : 0u, vb = b ? *b : 0u, vc = c ? *c : 0u;
		dst[0] = va; dst[1] = vb; dst[2] = vc;
		if (sum3_dev) { sum3_dev[0] = (double)va; sum3_dev[1] = (double)vb; sum3_dev[2] = (double)__uint_as_float(vc); }
		for (uint32_t k = 0; k < n_zero_words; ++k) zero_words[k] = 0u;
		__threadfence_system();
		__hip_atomic_store(&dst[3], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

int ngp_hip_post_words(void* stream, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t tag, uint32_t* dst4, uint32_t* zero_words, uint32_t n_zero_words, double* sum3_dev) {
	hipLaunchKernelGGL(post_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c, tag, dst4, zero_words, zero_words ? n_zero_words : 0u, sum3_dev);
	NGP_LAUNCH_CHECK("post_words_kernel");
	return 0;
}

int ngp_hip_gather_words(void* stream, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* dst4) {
	hipLaunchKernelGGL(gather_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c, d, dst4);
	NGP_LAUNCH_CHECK("gather_words_kernel");
	return 0;
}

int ngp_hip_grid_to_bitfield_and_pool(void* stream, const float* grid, uint32_t n_cascades_used, const float* mean_density, uint8_t* bitfield) {
	const uint32_t n = NGP_NERF_GRID_N_CELLS;
	hipLaunchKernelGGL(grid_to_bitfield_kernel, dim3(div_up(n / 8 * NGP_NERF_CASCADES, 256)), dim3(256), 0, (hipStream_t)stream, n / 8 * NGP_NERF_CASCADES, n / 8 * n_cascades_used, grid, bitfield, mean_density);
	NGP_LAUNCH_CHECK("grid_to_bitfield_kernel");
	for (uint32_t level = 1; level < NGP_NERF_CASCADES; ++level) {
		hipLaunchKernelGGL(bitfield_max_pool_kernel, dim3(div_up(n / 64, 256)), dim3(256), 0, (hipStream_t)stream, n / 64, bitfield + (size_t)(n / 8) * (level - 1), bitfield + (size_t)(n / 8) * level);
		NGP_LAUNCH_CHECK("bitfield_max_pool_kernel");
	}
	return 0;
}

} // extern "C"
