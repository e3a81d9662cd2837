/*
 * ngp_hip.h — C ABI of libngp_hip.so: the MI355X (gfx950) kernels behind the blender-ngp NeRF hot path.
 *
 * The reference has no C ABI: its only FFI seam is the pybind11 module `pyngp` (src/python_api.cu:306-888) and
 * everything below is C++ templates dispatching to CUDA / tiny-cuda-nn.  This header is the seam the north star asks
 * for ("host code stays C++ calling HIP through a thin C-ABI"): one entry point per reference kernel / tcnn call on
 * the path, cited below.  The host-side `Testbed` (blender-ngp_amd/host) is the only intended caller; INTEGRATION.md
 * shows how a maintainer of the reference would bind it.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all calls are stream-ordered, none syncs;
 *   - no allocation happens inside the library; scratch is passed in explicitly;
 *   - return value: 0 on success, otherwise a hipError_t (or a negative library code); ngp_hip_last_error() returns text;
 *   - fp16 data crosses the boundary as uint16_t (IEEE binary16 bits);
 *   - matrices are column-major like Eigen (3x4 camera matrix = 12 floats, column c at [3c..3c+2]).
 */
#ifndef NGP_HIP_H
#define NGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGP_HIP_ABI_VERSION 1

/* ---- constants (src/testbed_nerf.cu:53-73, include/neural-graphics-primitives/nerf.h:24-26) ---- */
#define NGP_NERF_GRIDSIZE 128u
#define NGP_NERF_GRID_N_CELLS (128u * 128u * 128u)
#define NGP_NERF_CASCADES 8u
#define NGP_NERF_STEPS 1024u
#define NGP_N_MAX_RANDOM_SAMPLES_PER_RAY 8u
#define NGP_GRIDMLP_N_PARAMS 7168u /* plumbing configs: one MLP 32->64->64->16 (configs/image/base.json, configs/sdf/base.json) */
#define NGP_MLP_N_PARAMS 10240u /* density 32->64->16 (3072) + rgb 32->64->64->16 (7168): configs/nerf/base.json:30-36,52-58 */

/* ---- PODs ---- */
typedef struct { float min[3], max[3]; } NgpAabb;                    /* bounding_box.cuh:43-268 */
typedef struct { float o[3], d[3]; } NgpRay;                         /* common.h:169-172 */
typedef struct { float start[12], end[12]; } NgpXForm;               /* common.h:174-177 TrainingXForm */
typedef struct { float pos[3]; float dt; float dir[3]; } NgpCoord;   /* nerf.h:86-107 NerfCoordinate (28 B) */
typedef struct {                                                     /* nerf.h:28-36 NerfPayload (40 B) */
	float origin[3]; float dir[3]; float t; float max_weight; uint32_t idx; uint16_t n_steps; uint8_t alive; uint8_t pad_;
} NgpPayload;
#define NGP_RAY_PAUSED 2 /* NgpPayload.alive between ngp_hip_generate_next_inputs (max_skips_per_pass > 0) and ngp_hip_composite; not in the reference */
typedef struct {                                                     /* nerf_loader.h:30-45 TrainingImageMetadata */
	const void* pixels;       /* device: RGBA8 (1), half4 (2), float4 (3) — common_device.cuh:621-626 */
	int32_t image_data_type;
	int32_t res[2];
	float focal_length[2];
	float principal_point[2];
	float rolling_shutter[4];
	int32_t lens_mode;        /* common.h:179-184: 0 perspective, 1 opencv, 2 ftheta, 3 latlong */
	float lens_params[7];
	const float* depth;       /* device or NULL */
	const NgpRay* rays;       /* device or NULL */
} NgpImageMeta;
typedef struct {                                                     /* render_request.cuh:55-103 RenderCameraProperties */
	float transform[12];      /* 3x4 column-major camera-to-world */
	int32_t model;            /* camera_models.cuh:27-31: 0 Perspective, 1 QuadrilateralHexahedron, 2 SphericalQuadrilateral */
	float focal_length;
	float sq_width, sq_height, sq_curvature;          /* SphericalQuadrilateral */
	float qh_front[12], qh_back[12];                  /* QuadrilateralHexahedron: tl, tr, bl, br of each face */
	float near_distance, aperture_size, focus_z;
} NgpRenderCamera;
typedef struct {                                                     /* testbed.h:603-616 ErrorMap CDFs as the kernels take them (testbed_nerf.cu:3243-3245) */
	const float* cdf_x_cond_y; /* device [n_images][res[1]][res[0]] or NULL (sample_focal_plane_proportional_to_error off) */
	const float* cdf_y;        /* device [n_images][res[1]]         or NULL (same switch) */
	const float* cdf_img;      /* device [n_images]                 or NULL (sample_image_proportional_to_error off) */
	int32_t res[2];
} NgpErrorMapCdf;
typedef struct { float scale; uint32_t resolution; uint32_t offset; uint32_t size; } NgpGridLevel;
typedef struct {                                                     /* tcnn GridEncoding geometry (HashGrid, F=2, 3-D) */
	uint32_t n_levels;        /* must be 16 (configs/nerf/base.json:23-29) */
	uint32_t n_grid_entries;  /* sum of level sizes; grid params = 2 * n_grid_entries */
	NgpGridLevel levels[16];
} NgpNetDesc;

/* enums mirror include/neural-graphics-primitives/common.h:103-141 */
enum { NGP_LOSS_L2 = 0, NGP_LOSS_L1 = 1, NGP_LOSS_MAPE = 2, NGP_LOSS_SMAPE = 3, NGP_LOSS_HUBER = 4, NGP_LOSS_LOG_L1 = 5, NGP_LOSS_RELATIVE_L2 = 6 };
enum { NGP_ACT_NONE = 0, NGP_ACT_RELU = 1, NGP_ACT_LOGISTIC = 2, NGP_ACT_EXPONENTIAL = 3 };
enum { NGP_COLOR_LINEAR = 0, NGP_COLOR_SRGB = 1 };
enum { NGP_TONEMAP_IDENTITY = 0, NGP_TONEMAP_ACES = 1, NGP_TONEMAP_HABLE = 2, NGP_TONEMAP_REINHARD = 3 };

int ngp_hip_abi_version(void);
const char* ngp_hip_last_error(void);

/* ============================ network (tiny-cuda-nn replacement) ============================ */

/* tcnn GridEncoding ctor (level scale / resolution / offset table); per_level_scale per src/testbed.cu:2313-2325. Host only. */
int ngp_hip_net_make_desc_host(uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale, NgpNetDesc* desc_host);
/* number of parameters of the base family: 10240 + 2*n_grid_entries; order density MLP, rgb MLP, grid (nerf_network.h:361-394). Host only. */
uint32_t ngp_hip_net_n_params_host(const NgpNetDesc* desc_host);

/* NerfNetwork as the reference BUILDS it from other configs (src/testbed.cu:2337-2363; nerf_network.h:76-101).  Every network entry point below takes an optional
 * trailing `const NgpNetVariant*`; NULL (or {0 extra dims, 2 hidden colour layers}) is configs/nerf/base.json — the fused MFMA kernels of csrc/network.hip.  Anything else
 * runs the same scheme with the layer list as a template parameter (csrc/network_netx_mfma.cuh: one more K-block for the extra dims, 0 / 1 / 3 hidden colour layers; MFMA
 * forward, fused dgrad + wgrad backward, dL_dinput for the camera-side trainables), or — flags & NGP_NETX_SCALAR — the scalar kernels of csrc/network_generic.cuh, which
 * are bit-compatible with oracle/orc_netx.c and serve as the checker:
 *   n_extra_dims 1..16     per-image latent codes (`n_extra_learnable_dims`) or light directions (`driver_parameters`): an Identity-encoded block behind the SH
 *                          block of the direction encoding (configs/nerf/base.json:37-51) — the colour network's input grows from 32 to 48;
 *   n_rgb_hidden_layers    0..3: configs/nerf/base_{0,1,2,3}layer.json (0 = one [16][in] matrix, no activation; h >= 1: [64][in], (h - 1) x [64][64], [16][64]).
 * Parameter order: density MLP [64][32], [16][64]; the colour matrices in layer order; the grid.  Host struct, device pointers. */
typedef struct {
	uint32_t n_extra_dims, n_rgb_hidden_layers;
	const float* extra_dims;       /* [rows][n_extra_dims] fp32 (Testbed::Nerf::Training::extra_dims_gpu, src/testbed_nerf.cu:2297-2318) or NULL */
	const uint32_t* sample_slot;   /* [n]: the row of extra_dims each sample uses (its ray's image, :1136); NULL: row 0 for all (rendering: get_inference_extra_dims, :2320-2337) */
	float* dL_dextra;              /* backward only, or NULL: [n][n_extra_dims] fp32 = the extra-dim rows of the network's dL_dinput (coords_gradient(j)->get_extra_dims(), :1741) */
	uint32_t flags;                /* NGP_NETX_SCALAR: run the scalar checker kernels (csrc/network_generic.cuh) instead of the MFMA kernels (csrc/network_netx_mfma.cuh) */
} NgpNetVariant;
enum { NGP_NETX_SCALAR = 1 };
uint32_t ngp_hip_net_mlp_params_host(const NgpNetVariant* variant);   /* 10240 for NULL / the base family; the grid follows behind */

/* NerfNetwork::initialize_params (nerf_network.h:396-441) driven by Trainer(seed) (src/testbed.cu:2445): fills the fp32 master
 * copy and the fp16 training + inference copies. */
int ngp_hip_nerf_init_params(void* stream, const NgpNetDesc* desc_host, uint64_t seed, float* master, uint16_t* params, uint16_t* inference_params, const NgpNetVariant* variant);

/* NerfNetwork::inference_mixed_precision_impl (nerf_network.h:103-137); call sites src/testbed_nerf.cu:2223, 3256,
 * src/nerf_renderer.cu:763.  coords: n records of `coord_stride_floats` floats (pos at 0..2, dir at 4..6).
 * out: fp16, sample i at out[i*out_stride + 0..3] = (r, g, b, sigma) raw network outputs.  `desc_dev` is a device copy of the desc. */
int ngp_hip_nerf_inference(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t coord_stride_floats,
                           uint32_t n, uint16_t* out, uint32_t out_stride, const NgpNetVariant* variant);

/* NerfNetwork::density (nerf_network.h:268-284); call site src/testbed_nerf.cu:2833.  out0[i] = density-net output channel 0 (fp16). */
int ngp_hip_nerf_density(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* pos, uint32_t pos_stride_floats,
                         uint32_t n, uint16_t* out0, const NgpNetVariant* variant);

/* NerfNetwork::forward_impl (nerf_network.h:143-185); call site src/testbed_nerf.cu:3330.  Same outputs as inference plus the
 * encoded features x_saved [n][32] fp16 that backward consumes (the tcnn ForwardContext). */
int ngp_hip_nerf_forward(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t coord_stride_floats,
                         uint32_t n, uint16_t* out, uint32_t out_stride, uint16_t* x_saved, const NgpNetVariant* variant);
/* Two-kernel variants of the three forward passes above (same results, bit for bit): an XCD-affine hash-encode kernel writes the
 * 32 features of every sample into level planes inside `workspace` (each XCD of the MI355X walks at most two 2-MiB level tables,
 * so its 4-MiB L2 holds them), then the MLP kernel reads the planes.  Faster whenever n is large enough to fill the chip
 * (training, occupancy-grid update, render passes); the single-kernel entry points need no workspace. */
uint64_t ngp_hip_nerf_encode_workspace_bytes(uint32_t n);
int ngp_hip_nerf_inference_ws(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t coord_stride_floats,
                              uint32_t n, uint16_t* out, uint32_t out_stride, void* workspace, uint64_t workspace_bytes, const NgpNetVariant* variant);
int ngp_hip_nerf_density_ws(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* pos, uint32_t pos_stride_floats,
                            uint32_t n, uint16_t* out0, void* workspace, uint64_t workspace_bytes, const NgpNetVariant* variant);
int ngp_hip_nerf_forward_ws(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, const float* coords, uint32_t coord_stride_floats,
                            uint32_t n, uint16_t* out, uint32_t out_stride, uint16_t* x_saved, void* workspace, uint64_t workspace_bytes, const NgpNetVariant* variant);

/* bytes of scratch ngp_hip_nerf_backward needs for a batch of n (n must be a multiple of 256). Host only.
 *   ..._bytes(n)                  enough for ANY level table (every level priced as a dense one: 160 bytes of sort records per sample and level);
 *   ..._bytes_for(desc_host, n)   what THIS level table needs: the records of the 16 levels are packed, a hashed level takes 48 bytes per sample, a dense one 160
 *                                 (configs/nerf/base.json at n = 2^18: 405 MB instead of 778 MB).  A caller that sizes its scratch this way passes the same
 *                                 `desc_host` to ngp_hip_nerf_backward, and `desc_dev` must be a device copy of exactly that struct: host and device derive the
 *                                 record offsets from their own copy (desc_host == NULL: the size check falls back to ..._bytes(n)). */
uint64_t ngp_hip_nerf_backward_scratch_bytes(uint32_t n);
uint64_t ngp_hip_nerf_backward_scratch_bytes_for(const NgpNetDesc* desc_host, uint32_t n);

/* NerfNetwork::backward_impl (nerf_network.h:187-266) with EGradientMode::Overwrite; call site src/testbed_nerf.cu:3331.
 * dL_dout: fp16 [n][dl_stride] with channels 0..3 consumed (extract_rgb 46-60, add_density_gradient 63-74).
 * grads: fp16 [n_params]; the whole vector is overwritten (every entry written exactly once: no memset, no global float atomics).
 * Optional trailing arguments (each may be NULL):
 *   mlp_done_event        a caller-owned hipEvent_t recorded right behind the fused MLP kernel (dgrad + weight gradients; the 256-register kernel of the pass) — a host
 *                         that runs other work next to the backward can hold that work back until this kernel is through;
 *   grid_gradients_event  recorded once all of `grads` is final (the hash-grid part — everything behind the MLP parameters — last): a data-parallel host starts
 *                         its exchange on another stream at that point;
 *   dL_dinput             fp32 [n][6]: the gradient with respect to the network INPUT — NerfNetwork::backward_impl with a dL_dinput matrix, which the training step
 *                         requests when camera parameters train (prepare_input_gradients, src/testbed_nerf.cu:3324-3346): d/d(pos x, y, z) through the hash encoding
 *                         ([tcnn] GridEncoding backward to the input: fp32 sum over levels and features of dL/dy * dy/dx of the trilinear interpolation) and
 *                         d/d(dir x, y, z) through the SH basis, both in the warped [0, 1] coordinates of NgpCoord; dt carries no gradient (base family only);
 *   variant               see NgpNetVariant (its dL_dextra receives the extra-dim gradients).
 * desc_host (may be NULL) is the host copy of the level table `desc_dev` points to: the host side lays out the sort's record space and the owners' launch grid from it
 * while the kernels read desc_dev.  The first call with a new (desc_dev, contents of desc_host) pair reads desc_dev back once (stream-ordered; the host waits) and refuses a
 * desc_host that is a different table; NULL = every level priced as the worst case, whole owner grid launched. */
int ngp_hip_nerf_backward(void* stream, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host, const uint16_t* params, const float* coords,
                          uint32_t coord_stride_floats, uint32_t n, const uint16_t* x_saved, const uint16_t* dL_dout, uint32_t dl_stride,
                          uint16_t* grads, void* scratch, uint64_t scratch_bytes, void* mlp_done_event, void* grid_gradients_event, float* dL_dinput, const NgpNetVariant* variant,
                          const uint32_t* x_row_index /* NULL: row k of x_saved belongs to sample k.  Else (base network family): sample k's encoding is row x_row_index[k] of
                          x_saved — the UNCOMPACTED batch's rows, with the index list ngp_hip_compute_loss left (NgpLossExtras::x_row_index_out) instead of copies of the rows */);
/* The backward pass over the LIVE samples of a batch.  A sample whose loss gradient is zero in all four channels — after the loss kernel and fill_rollover_and_rescale
 * (src/testbed_nerf.cu:1280-1597, 3314-3322), in fp16: the tails of the rays, where the transmittance has not yet reached the 1e-4 cut but weight x loss_scale has left the
 * fp16 range; 30-45 % of a training batch — adds exact zeros to every sum NerfNetwork::backward_impl forms.
 *   ngp_hip_compact_live_samples   lists the other samples: live_index_out[k] = the row of the k-th live sample (within a workgroup of 256 rows in order, the workgroups in
 *                                  arrival order), coords_out = their coordinate rows next to each other (same stride), *n_live_dev += their number (must be 0 on entry);
 *   ngp_hip_nerf_backward_live     ngp_hip_nerf_backward (base network family, no dL_dinput) over those samples: the MFMA kernel reads rows live_index[k] of coords / x_saved /
 *                                  dL_dout for k < *n_live_dev, the binning passes of the hash-grid backward read coords_live and stop behind the live samples (the wave lanes
 *                                  that would have carried zeros are gone).  zero_word_dev (may be NULL): a device word the first kernel clears — the counter the NEXT step's
 *                                  compaction adds to.
 * Hash-grid gradients: the same bits as the whole batch gives (exact sums; zeros change nothing).  MLP weight gradients: the same sums with another association of the fp32
 * adds (inside the oracle tolerance of ngp_hip_nerf_backward). */
int ngp_hip_compact_live_samples(void* stream, uint32_t n, const uint16_t* dL_dout, uint32_t dl_stride, const float* coords, uint32_t coord_stride_floats,
                                 uint32_t* live_index_out, float* coords_out, uint32_t* n_live_dev);
int ngp_hip_nerf_backward_live(void* stream, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host, const uint16_t* params, const float* coords,
                               uint32_t coord_stride_floats, uint32_t n, const uint16_t* x_saved, const uint16_t* dL_dout, uint32_t dl_stride,
                               uint16_t* grads, void* scratch, uint64_t scratch_bytes, void* mlp_done_event, void* grid_gradients_event,
                               const uint32_t* live_index, const float* coords_live, const uint32_t* n_live_dev, uint32_t* zero_word_dev,
                               const uint32_t* x_row_index /* as in ngp_hip_nerf_backward: applied to the row live_index[k] names */);
/* Per-image extra dims of a training step.  ray_image[i] = the image of kept ray i (what image_idx gives the ray generator, src/testbed_nerf.cu:1131-1136);
 * numsteps = (count, base) pairs of the kept rays (the ray generator's before compaction, the loss kernel's after).
 *   ngp_hip_ray_images            ray_image[i] = image_idx(ray_indices[i], ...) for the kept rays (:1062-1083);
 *   ngp_hip_expand_ray_slots      sample_slot[base + j] = ray_image[i] for every sample of every kept ray (the reference appends the image's row to every coordinate, :1246);
 *   ngp_hip_rollover_slots        sample_slot[k] = sample_slot[k % *n_input_elements] for the padded tail of a compacted batch (fill_rollover, :3314-3322);
 *   ngp_hip_extra_dims_gradient   compute_extra_dims_gradient_train_nerf (:1710-1746): gradient[ray_image[i]][k] += sum over the ray's compacted samples of dL_dextra. */
int ngp_hip_ray_images(void* stream, uint32_t n_rays_capacity, const uint32_t* rays_counter, const uint32_t* ray_indices, uint32_t n_rays_global, uint32_t n_training_images,
                       const float* cdf_img /* NgpErrorMapCdf.cdf_img or NULL */, uint32_t* ray_image);
int ngp_hip_expand_ray_slots(void* stream, uint32_t n_rays_capacity, const uint32_t* rays_counter, const uint32_t* ray_image, const uint32_t* numsteps, uint32_t n_samples_capacity, uint32_t* sample_slot);
int ngp_hip_rollover_slots(void* stream, uint32_t n_elements, const uint32_t* n_input_elements, uint32_t* sample_slot);
int ngp_hip_extra_dims_gradient(void* stream, uint32_t n_rays_capacity, const uint32_t* rays_counter, const uint32_t* ray_image, const uint32_t* numsteps, const float* dL_dextra, uint32_t n_extra_dims,
                                float* gradient);

/* Trainer::optimizer_step(stream, loss_scale) (src/testbed_nerf.cu:2950) with Ema{decay} o ExponentialDecay o Adam as configured by
 * configs/nerf/base.json:5-22.  `step` = 1-based optimizer step; `learning_rate` = base lr after ExponentialDecay (host applies it). */
/* optimize_mask: tcnn Adam's `optimize_matrix_params` (bit 0: the MLP weights) and `optimize_non_matrix_params` (bit 1: the
 * encoding) switches — what Testbed::train sets from `shall_train_network` / `shall_train_encoding` every step (src/testbed.cu:2556-2563).  A
 * parameter class that is switched off keeps its weights and moments; the Ema copy is updated for every parameter either way. */
int ngp_hip_optimizer_step(void* stream, uint32_t n_params, uint32_t n_matrix_params, uint32_t step, float learning_rate, float beta1, float beta2,
                           float epsilon, float l2_reg, float loss_scale, float ema_decay, const uint16_t* grads, float* master, uint16_t* params,
                           float* first_moments, float* second_moments, float* ema, uint16_t* inference_params, uint32_t optimize_mask /* 3: the whole step */);
/* Two more bits of optimize_mask split the step into its stages, for the SHARDED optimizer step of the data-parallel path: a rank runs the Adam stage
 * on its 1 / world of the parameters (pointers advanced to the shard, n_matrix_params counted from there), the fp16 weights are all-gathered, and the Ema stage — which
 * reads the fp16 weights only — runs over all parameters on every rank.  Element-wise the same arithmetic as the unsplit step, bit for bit. */
#define NGP_OPT_NO_EMA 4u     /* Adam stage alone: ema / inference_params are not touched (may be NULL) */
#define NGP_OPT_EMA_ONLY 8u   /* Ema stage alone: grads / master / moments are not touched (may be NULL) */
/* fp16 <-> fp32 copies around the fp32 reduce-scatter of that step: dst[i] = src[i] for i < n, 0 for n <= i < n_padded */
int ngp_hip_f16_to_f32(void* stream, uint32_t n, uint32_t n_padded, const uint16_t* src, float* dst);
int ngp_hip_f32_to_f16(void* stream, uint32_t n, const float* src, uint16_t* dst);

/* ============================ occupancy grid (src/testbed_nerf.cu:369-610, 2761-2859) ============================ */
int ngp_hip_mark_untrained_density_grid(void* stream, uint32_t n_elements, float* grid_out, uint32_t n_training_images,
                                        const NgpImageMeta* metadata, const NgpXForm* xforms, int clear_visible_voxels);      /* :369 */
int ngp_hip_generate_grid_samples_nonuniform(void* stream, uint32_t n_elements, uint64_t rng_state, uint64_t rng_inc, uint32_t step,
                                             const NgpAabb* aabb_host, const float* grid_in, float* out_pos, uint32_t* indices,
                                             uint32_t n_cascades, float thresh);                                              /* :465 */
/* Not in the reference: the SAME samples as the call above — the same (position, index) pairs, each produced by the reference's body for its sample number i
 * (:465-494) — but written in another order: thread c stands on cell c (Morton order), inverts the first try's index map for i, and the samples of a workgroup's 1024 cells take
 * one contiguous slot range in cell order.  Neighbouring slots then hold neighbouring cells, so the density pass (:2833) and the splat (:496) that follow touch shared lines from
 * neighbouring lanes.  Order-independent consumers (max-splat) give the same grid bit for bit.  No atomics: the slot of a sample is a function of (n, step).
 * workspace: ngp_hip_generate_grid_samples_morton_workspace_bytes() of device scratch; n_elements <= 2^24. */
uint64_t ngp_hip_generate_grid_samples_morton_workspace_bytes(void);
int ngp_hip_generate_grid_samples_morton(void* stream, uint32_t n_elements, uint64_t rng_state, uint64_t rng_inc, uint32_t step,
                                         const NgpAabb* aabb_host, const float* grid_in, float* out_pos, uint32_t* indices,
                                         uint32_t n_cascades, float thresh, uint32_t* workspace);
int ngp_hip_splat_grid_samples_max(void* stream, uint32_t n_elements, const uint32_t* indices, const uint16_t* network_output,
                                   float* grid_out, int density_activation);                                                  /* :496 */
int ngp_hip_ema_grid_samples(void* stream, uint32_t n_elements, float decay, float* grid_out, const float* grid_in);          /* :532 */
/* reduce_sum(max(v,0)/n) over cascade 0 (:2851-2852): writes one float to mean_out (zeroed inside). */
int ngp_hip_density_grid_mean(void* stream, const float* grid, uint32_t n_elements, float* mean_out);
/* grid_to_bitfield (:563) + the 7 bitfield_max_pool levels (:589) of update_density_grid_mean_and_bitfield (:2854-2858): one launch per level that has cells of its own
 * (n_cascades_used), ONE launch for all the levels above (pure pools of the last cascade).  Same bytes as the reference's seven launches. */
int ngp_hip_grid_to_bitfield_and_pool(void* stream, const float* grid, uint32_t n_cascades_used, const float* mean_density, uint8_t* bitfield);
/* The tail of update_density_grid_nerf in one call — ema (:2838 -> :532), mean of cascade 0 (:2851-2852), bitfield + pooled levels (:2854-2858): the results of
 * ngp_hip_ema_grid_samples, ngp_hip_density_grid_mean (to the summation order; here a fixed one) and ngp_hip_grid_to_bitfield_and_pool, in two passes over the grid
 * instead of a memset and three.  grid_out / grid_in: n_cascades_used x 2^21 cells.  workspace: ngp_hip_density_grid_tail_workspace_bytes() of scratch. */
uint64_t ngp_hip_density_grid_tail_workspace_bytes(void);
int ngp_hip_density_grid_ema_mean_bitfield(void* stream, uint32_t n_cascades_used, float decay, float* grid_out, const float* grid_in, float* mean_out, uint8_t* bitfield, void* workspace);
/* Not in the reference: one bit per 4x4x4 brick of cascade 0 (Morton order: 64 consecutive cells) that is set when any of its cells is
 * occupied — 1024 words.  The training march answers most of its empty-space lookups from this summary; it builds it itself from the
 * bitfield when none is passed, which costs every workgroup a pass over the 256 KB of cascade 0. */
int ngp_hip_bitfield_brick_summary(void* stream, const uint8_t* bitfield, uint32_t* summary_out_1024_words);

/* ============================ training rays (src/testbed_nerf.cu:1085-1260) ============================ */
/* generate_training_samples_nerf.  ray_offset / n_rays_global are the data-parallel extension: thread i marches global ray
 * ray_offset+i out of n_rays_global (image choice, rng stream); pass (0, n_rays) for the reference's single-GPU behaviour. */
/* march_mode: the marching kernel (same rays, same samples, bit for bit; slot order aside):
 *   NGP_MARCH_AUTO          wave-per-ray, all workgroups at once
 *   NGP_MARCH_LANE_PER_RAY  one lane per ray + a wave-per-ray expansion kernel: a latency-bound serial chain (~330 us at 2^14 rays) that costs
 *                           few issue slots (rounds 1-3: the only kernels for cone stepping)
 *   NGP_MARCH_WAVE_PER_RAY  64 step candidates per wave at once — on the closed-form step sequence when cone_angle_constant == 0 (~110 us on its own),
 *                           on the sequence the ray's own lane generates into LDS first when it is not (cone stepping: every aabb_scale > 1 dataset);
 *                           the one to run IN stream order
 *   NGP_MARCH_WAVE_PER_RAY_SHARED  the same kernels on 2.5 persistent workgroups per CU: the one to run NEXT TO the backward pass */
enum { NGP_MARCH_AUTO = 0, NGP_MARCH_LANE_PER_RAY = 1, NGP_MARCH_WAVE_PER_RAY = 2, NGP_MARCH_WAVE_PER_RAY_SHARED = 3 };
int ngp_hip_generate_training_samples(
	void* stream, uint32_t n_rays, const NgpAabb* aabb_host, uint32_t max_samples, uint64_t rng_state, uint64_t rng_inc,
	uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, NgpRay* rays_out_unnormalized, uint32_t* numsteps_out,
	NgpCoord* coords_out, uint32_t n_training_images, const NgpImageMeta* metadata, const NgpXForm* xforms, const uint8_t* density_grid,
	int max_level_rand_training, float* max_level_ptr, int snap_to_pixel_centers, int train_envmap, float cone_angle_constant,
	const float* distortion_data, const int32_t* distortion_resolution_host, uint32_t ray_offset, uint32_t n_rays_global,
	const NgpErrorMapCdf* cdf_host /* NULL: uniform image / pixel choice */,
	const uint32_t* brick_summary /* NULL, or what ngp_hip_bitfield_brick_summary wrote for density_grid (same samples either way) */, uint32_t march_mode);

/* ============================ load-time image sharpening (src/nerf_loader.cu:102-123, 803-825) ============================
 * NerfDataset::set_training_image with sharpen_amount > 0: Byte images first become premultiplied linear half4 (from_rgba32<__half>,
 * common_device.cuh:562-590; mask_color pixels -> -1), then the 5-tap unsharp filter with centre weight 4 + 1/amount runs out of place
 * (neighbours wrap around the image like in the reference).  image_data_type: 2 half4, 3 float4. */
int ngp_hip_image_from_rgba32_f16(void* stream, uint64_t n_pixels, const uint8_t* rgba8, uint16_t* out_half4, uint32_t mask_color);
int ngp_hip_image_sharpen(void* stream, uint64_t n_pixels, uint32_t width, const void* pix, void* dest, int image_data_type, float sharpen_amount);

/* ============================ error-map CDFs (src/testbed_nerf.cu:1982-2037, 2971-3024) ============================ */
/* construct_cdf_2d: per image and row, running sum of (error + 1e-10) over x, normalised and blended with MIN_PDF = 0.01 of uniform;
 * cdf_y[img][y] receives the row totals.  construct_cdf_1d: the same over the row totals, cdf_img[img] receives the image total (the
 * host turns those into the image CDF with MIN_PMF = 0.1, :3000-3015).  Both overwrite their outputs. */
int ngp_hip_construct_cdf_2d(void* stream, uint32_t n_images, uint32_t height, uint32_t width, const float* data, float* cdf_x_cond_y, float* cdf_y);
int ngp_hip_construct_cdf_1d(void* stream, uint32_t n_images, uint32_t height, float* cdf_y, float* cdf_img);

/* The loss kernel with the environment map (compute_loss_kernel_train_nerf's envmap_data / envmap_gradient / envmap_resolution / envmap_loss_type,
 * :1289-1292): the map (fp32 rgba [h][w][4], TrainableBuffer<4,2,float>::params) is composited in front of the background colour of every ray
 * (:1394-1401); with envmap_gradient (train_envmap; the caller clears it every step, :2941-2944) rays whose every sample was kept deposit
 * loss_scale * T * dL/drgb [/ srgb'(background)] bilinearly (:1573-1596, envmap.cuh:65-103; alpha gets no gradient).  extras_host NULL: no environment map, no sharpness weighting. */
typedef struct {
	const float* envmap_data; float* envmap_gradient; int32_t envmap_res[2]; int32_t envmap_loss_type;
	/* include_sharpness_in_error (testbed.h:670; :1321-1323, 1476-1485): per image a sharpness_res grid of tile sharpness (ngp_hip_compute_sharpness) and a cascaded
	 * 128^3 grid of the sharpest tile that has seen each cell (decayed by ngp_hip_decay_grid every training_prep, :2901-2912); the error a ray deposits into the error
	 * map is scaled by max(sharp / grid_sharp, 0.01).  NULL: off */
	const float* sharpness_data; int32_t sharpness_res[2]; float* sharpness_grid;
	/* Instead of carrying encoding rows through the compaction (encoded_in / encoded_out of ngp_hip_compute_loss, 128 bytes of traffic per kept sample): slot k of the
	 * compacted batch gets the index of the uncompacted sample it came from ([max_samples_compacted] words; roll it over like the coordinates, stride 1), and
	 * ngp_hip_nerf_backward reads row x_row_index[k] of the UNCOMPACTED x_saved.  NULL: off */
	uint32_t* x_row_index_out;
} NgpLossExtras;
/* ============================ loss + compaction (src/testbed_nerf.cu:1280-1597, 3314-3322) ============================
 * Forward pass.  The reference runs inference on all samples (:3256), compacts, then runs m_network->forward on the compacted batch
 * (:3330) only to rebuild the activations backward needs; the network outputs of that second pass are never read.  ngp_hip_nerf_backward
 * recomputes the MLPs from the saved encoding, so all it needs from "forward" is x_saved — a pure function of the sample position.  A
 * host may therefore run ngp_hip_nerf_forward instead of ngp_hip_nerf_inference on the uncompacted samples and let ngp_hip_compute_loss
 * carry each kept sample's 64-byte encoding row along with its coordinates (encoded_in / encoded_out), then roll it over like the
 * coordinates (ngp_hip_fill_rollover_f32 with stride 16): same bits as the second pass would produce, one gather pass less.
 * loss_output ([n_rays] floats, optional): every slot is written — the ray's loss / n_rays, or 0 for slots that hold no ray or whose ray
 * got no room in the compacted batch (the reference leaves those at the zero its per-step memset put there, :2864). */
int ngp_hip_compute_loss(
	void* stream, uint32_t n_rays, const NgpAabb* aabb_host, uint64_t rng_state, uint64_t rng_inc, uint32_t max_samples_compacted,
	const uint32_t* rays_counter, float loss_scale, uint32_t mlp_stride, const float* background_color_host, int color_space,
	int train_with_random_bg_color, int train_in_linear_colors, uint32_t n_training_images, const NgpImageMeta* metadata,
	const uint16_t* network_output, uint32_t* numsteps_counter, const uint32_t* ray_indices_in, const NgpRay* rays_in_unnormalized,
	uint32_t* numsteps_in, const NgpCoord* coords_in, NgpCoord* coords_out, uint16_t* dloss_doutput, uint32_t dl_stride, int loss_type,
	float* loss_output, int max_level_rand_training, float* max_level_compacted, int rgb_activation, int density_activation,
	int snap_to_pixel_centers, float* error_map, const int32_t* error_map_res_host, const float* mean_density, const float* exposure,
	float near_distance, const NgpErrorMapCdf* cdf_host /* NULL: uniform; must be what ngp_hip_generate_training_samples got */,
	const uint16_t* encoded_in, uint16_t* encoded_out /* both NULL, or: row i of encoded_in ([n_samples][32] fp16, the x_saved that
	ngp_hip_nerf_forward wrote for coords_in) is copied next to coords_out — see "forward pass" below */,
	float depth_supervision_lambda /* 0: off (testbed.h:680) */, int depth_loss_type /* ELossType for the depth term, reference default L1 (testbed.h:654);
	target = |ray.d| * metadata[img].depth at the ray's pixel, rays of images without depth are unaffected (:1450-1452, 1536-1541) */,
	float* exposure_gradient /* NULL, or [n_images][3] floats that receive (atomicAdd, the caller clears them) the gradient of the loss with
	respect to the per-image exposures (:1558-1572, optimize_exposure) */,
	const NgpLossExtras* extras_host /* NULL: no environment map, no sharpness weighting (NgpLossExtras above) */);
/* compute_cam_gradient_train_nerf (:1600-1712, call site :3350-3378), the extrinsics outputs: per kept ray, the network's input gradient of its COMPACTED samples
 * (coords_gradient: [sample][6] fp32 as ngp_hip_nerf_backward writes its dL_dinput; numsteps_in holds the compacted (count, base) pairs ngp_hip_compute_loss left) is
 * folded into a ray-origin and a ray-direction gradient and added (atomicAdd; the caller clears them every n_steps_between_cam_updates, :2916-2918) to
 * cam_pos_gradient[img] and, as the angle-axis ray.d x grad_d, to cam_rot_gradient[img] ([n_images][3] floats each; either may be NULL), both divided by the pixel
 * pdf of the ray's draw.  rng / cdf_host must be what ngp_hip_compute_loss got.  The reference kernel takes a cam_focal_length_gradient pointer and never writes it.
 * (declaration below, behind the sharpness helpers) */
/* compute_sharpness (src/nerf_loader.cu:129-169): sharpness_out[y][x] = variance of the Laplacian of the luma over tile (x, y) of a sharpness_res grid laid over the image */
int ngp_hip_compute_sharpness(void* stream, const int32_t* sharpness_res_host, const int32_t* image_res_host, const void* pixels, int image_data_type, float* sharpness_out);
int ngp_hip_decay_grid(void* stream, uint32_t n_elements, float decay, float* grid);   /* decay_sharpness_grid_nerf (:557-561) */
/* The lens-distortion branch (:1671-1685; xforms .. distortion_resolution_host, all NULL: off): the ray-direction gradient minus its component along the ray, rotated by
 * the inverse of the image's camera rotation (xforms[img].start), is splatted (x, y; divided by the pixel pdf) into distortion_gradient at the ray's pixel and
 * the bilinear weights into distortion_gradient_weight (both fp32 [h][w][2], atomicAdd; the caller clears them every n_steps_between_cam_updates, :2919-2920). */
int ngp_hip_compute_cam_gradient(
	void* stream, uint32_t n_rays, const NgpAabb* aabb_host, uint64_t rng_state, uint64_t rng_inc, const uint32_t* rays_counter, int snap_to_pixel_centers,
	float* cam_pos_gradient, float* cam_rot_gradient, uint32_t n_training_images, const NgpImageMeta* metadata, const uint32_t* ray_indices_in,
	const NgpRay* rays_in_unnormalized, const uint32_t* numsteps_in, const NgpCoord* coords_compacted, const float* coords_gradient, const NgpErrorMapCdf* cdf_host,
	const NgpXForm* xforms, float* distortion_gradient, float* distortion_gradient_weight, const int32_t* distortion_resolution_host);
/* safe_divide (:2039-2045, call site :3086-3090): inout[i] = divisor[i] > 0 ? inout[i] / divisor[i] : 0 */
int ngp_hip_safe_divide(void* stream, uint32_t n_elements, float* inout, const float* divisor);
/* Trainer<float, float, float>::optimizer_step(stream, loss_scale) of the envmap / distortion-map TrainableBuffers (:2955, :3091) with [Ema o] ExponentialDecay o Adam
 * (configs/nerf/base.json:59-101): all-fp32 Adam in which every parameter is a non-matrix parameter (zero gradient: skipped; no l2_reg), `learning_rate` = base rate
 * after ExponentialDecay (the host applies it), `step` 1-based, ema NULL or the fp32 Ema copy (ema_decay its decay).  In place. */
int ngp_hip_optimizer_step_f32(void* stream, uint32_t n_params, uint32_t step, float learning_rate, float beta1, float beta2, float epsilon, float loss_scale, float ema_decay,
                               const float* grads, float* params, float* first_moments, float* second_moments, float* ema);
/* tcnn fill_rollover_and_rescale<half> / fill_rollover<float> (call sites :3314-3322) */
int ngp_hip_fill_rollover_and_rescale_f16(void* stream, uint32_t n_elements, uint32_t stride, const uint32_t* n_input_elements, uint16_t* inout);
int ngp_hip_fill_rollover_f32(void* stream, uint32_t n_elements, uint32_t stride, const uint32_t* n_input_elements, float* inout);
/* The roll-overs of one training step in ONE launch: fill_rollover_and_rescale<half> on dloss, fill_rollover<float> on the compacted coordinates
 * (stride 7) and, optionally, on the carried encoding rows (stride 16 floats = 32 halves; see "Forward pass").  Element-wise the same arithmetic. */
int ngp_hip_fill_rollover_training(void* stream, uint32_t n_elements, const uint32_t* n_input_elements, uint16_t* dloss, uint32_t dl_stride, float* coords, uint32_t coord_stride_floats,
                                   float* encoded, uint32_t encoded_stride_floats);
/* ngp_hip_post_words (below) and ngp_hip_fill_rollover_training in one launch: the counters reach the polling host first, the roll-overs follow in the same kernel
 * (one launch less on the step's chain; only the positions that need a fill are visited).  dst4 NULL: roll-overs only. */
int ngp_hip_post_words_and_fill_rollover_training(void* stream, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t tag, uint32_t* dst4, uint32_t* zero_words, uint32_t n_zero_words,
                                                  double* sum3_dev, uint32_t n_elements, const uint32_t* n_input_elements, uint16_t* dloss, uint32_t dl_stride, float* coords,
                                                  uint32_t coord_stride_floats, float* encoded, uint32_t encoded_stride_floats);
/* tcnn reduce_sum(float*) as used for the loss scalar (:2887): sum of n floats into out (zeroed inside). */
int ngp_hip_reduce_sum_f32(void* stream, const float* in, uint32_t n, float* out);
/* NerfCounters::update_after_training (:2870-2874) reads its counters with blocking 4-byte copies.  This gathers up to four
 * 32-bit device words (NULL -> 0) into dst4[0..3] in stream order; dst4 may be device memory or host-mapped pinned memory, so the
 * host can pick the step's counters up from an event instead of draining the stream. */
int ngp_hip_gather_words(void* stream, const uint32_t* a, const uint32_t* b, const uint32_t* c, const uint32_t* d, uint32_t* dst4);
/* Same for a host that POLLS host-mapped memory instead of waiting on an event: dst4[0..2] = *a, *b, *c (NULL -> 0), then — after a
 * system-scope fence — dst4[3] = tag, so a reader that sees the tag sees the three words.  zero_words (optional) gets n_zero_words (a few) words
 * cleared in the same launch (the next step's march and compaction counters, which would otherwise cost memsets on the step's critical chain). */
int ngp_hip_post_words(void* stream, const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t tag, uint32_t* dst4, uint32_t* zero_words, uint32_t n_zero_words,
                       double* sum3_dev /* optional DEVICE copy {(double)*a, (double)*b, (double)(float)*c}: the operand of a data-parallel host's counter all-reduce */);

/* ============================ renderer (src/testbed_nerf.cu:612-989, 1748-1978; src/render_buffer.cu:235-348, 540-567) ============
 * NgpRenderExtras carries what init_rays_with_payload_kernel_nerf (:1809-1978) and composite_kernel_nerf (:767-989) take beyond their
 * camera / ray arguments, and the row range of a frame that is rendered in shards (SURVEY.md §8e "Render: image tiles/rows per rank + gather").
 * Host struct; the pointers inside are device pointers (or NULL).  A NULL NgpRenderExtras* means all defaults (whole frame, nothing extra). */
typedef struct {                                                     /* mask_3D.cuh:129-255 Mask3D (shared with the Blender renderer below) */
	int32_t mode;             /* EMaskMode: 0 Add, 1 Subtract */
	int32_t shape;            /* EMaskShape: 0 Box, 1 Cylinder, 2 Sphere, 3 All */
	float transform[16], itransform[16];
	float config[6];          /* box: dims xyz; cylinder: radius, height; sphere: radius */
	float feather, opacity;
} NgpMask3D;
typedef struct {
	const NgpMask3D* render_masks; uint32_t n_render_masks;   /* Testbed::prepare_nerf_masks (:2339-2352): rays that hit no mask die in init_rays (:1943-1956);
	                                                             per sample, weight *= clamp(1 + sum of mask.sample(pos), 0, 1) (:833-840) */
	int32_t glow_mode; float glow_y_cutoff;                   /* :843-939 */
	const float* envmap; int32_t envmap_res[2];               /* fp32 rgba [h][w][4] (TrainableBuffer<4,2,float>::params_inference); init_rays writes
	                                                             read_envmap(dir) into frame_buffer (:1931-1933; envmap.cuh:29-63) */
	const float* distortion; int32_t distortion_res[2];       /* fp32 [h][w][2] (TrainableBuffer<2,2,float>): added to the ray direction before the camera
	                                                             rotation (common_device.cuh:297-299); Distortion render mode paints it (:1959-1970) */
	int32_t quilting_dims[2];                                 /* {1,1} (or {0,0}): off; apply_quilting (common_device.cuh:541-560, :1853-1863) */
	int32_t render_mode;                                      /* ERenderMode of the frame (init_rays only needs to know Distortion = 5) */
	float* frame_buffer;                                      /* [h][w][4] fp32, written by init_rays for the envmap background and the Distortion mode */
	int32_t row_begin, row_end;                               /* {0,0}: the whole frame.  Else init_rays sets up the pixels of rows [row_begin, row_end) only:
	                                                             payloads[i] is pixel row_begin * width + i, payload.idx stays the pixel's index in the WHOLE frame (what
	                                                             shade writes to and every per-pixel random number is keyed by), so a frame cut into row ranges — one per
	                                                             rank — has the pixels of the frame rendered at once, bit for bit */
	int32_t tile_order;                                       /* 1: payload slots in 8 x 8 pixel tiles (slot = tile-linear) instead of row-major, so that a wave of the later
	                                                             kernels holds a square of neighbouring pixels; needs width and row count multiples of 8.  Same pixels. */
} NgpRenderExtras;
/* compact_kernel_nerf (:1784-1807) folded into the kernel that decides a ray's fate: with a NgpCompactOut, ngp_hip_advance_pos / ngp_hip_composite write every ray
 * ONCE — alive rays to dst_* at counter++, dead rays with alpha > 0.001 to dst_final_* at final_counter++, the others nowhere — instead of updating payloads / rgba /
 * depth in place for a ngp_hip_compact_rays pass to re-read.  The destination arrays must not alias the kernel's own arrays; the counters are bumped (the caller zeroes
 * `counter` per pass).  Ray order inside the compacted arrays differs from the separate pass (rays are independent: same pixels).  NULL: in place, as the reference. */
typedef struct {
	float* dst_rgba; float* dst_depth; NgpPayload* dst_payloads; float* dst_final_rgba; float* dst_final_depth; NgpPayload* dst_final_payloads;
	uint32_t* counter; uint32_t* final_counter;
	/* optional (both or neither): the workgroup that finishes last stores {*counter, sequence} to host_mailbox — two words, 8-byte aligned, of host memory the
	 * device can write (hipHostMalloc, coherent), written with ONE 8-byte store — so the caller can poll for `sequence` in the second word instead of copying
	 * *counter back behind a stream synchronisation.  blocks_done: a device word, zero before the first call; the kernel leaves it zero. */
	uint32_t* blocks_done; uint32_t* host_mailbox; uint32_t sequence;
} NgpCompactOut;
int ngp_hip_init_rays(void* stream, uint32_t sample_index, NgpPayload* payloads, const int32_t* res_host, const float* focal_length_host,
                      const float* camera_matrix0_host, const float* camera_matrix1_host, const float* rolling_shutter_host,
                      const float* screen_center_host, const float* parallax_shift_host, int snap_to_pixel_centers, const NgpAabb* render_aabb_host,
                      const float* render_aabb_to_local_host, float near_distance, int lens_mode, const float* lens_params_host,
                      float* depthbuffer, float plane_z /* focus distance; < 0: slice plane at -plane_z */, float aperture_size /* 0: pinhole */,
                      const NgpRenderCamera* camera_models_host /* NULL or model 0: Perspective; else only model / sq_* / qh_* are read (:1868-1908) */,
                      const NgpRenderExtras* extras_host);                                                                       /* :1809 */
/* (the start jitter is keyed by payload.idx, the pixel's index in the whole frame — the reference's ray index i, :625, when the frame is traced at once) */
int ngp_hip_advance_pos(void* stream, uint32_t n_elements, const NgpAabb* render_aabb_host, const float* render_aabb_to_local_host,
                        uint32_t sample_index, NgpPayload* payloads, const uint8_t* density_grid, uint32_t min_mip, float cone_angle_constant,
                        const NgpCompactOut* compact_host /* NULL: in place */,
                        const uint32_t* brick_summary /* NULL, or what ngp_hip_bitfield_brick_summary wrote for density_grid: empty-space lookups of cascade 0
                                                         are answered from a copy of it in LDS instead of global memory (same bits, same march) */); /* :612 */
int ngp_hip_compact_rays(void* stream, uint32_t n_elements, const float* src_rgba, const float* src_depth, const NgpPayload* src_payloads,
                         float* dst_rgba, float* dst_depth, NgpPayload* dst_payloads, float* dst_final_rgba, float* dst_final_depth,
                         NgpPayload* dst_final_payloads, uint32_t* counter, uint32_t* final_counter);                          /* :1784 */
int ngp_hip_generate_next_inputs(void* stream, uint32_t n_elements, const NgpAabb* render_aabb_host, const NgpAabb* train_aabb_host,
                                 NgpPayload* payloads, NgpCoord* network_input, uint32_t n_steps, const uint8_t* density_grid, uint32_t min_mip,
                                 float cone_angle_constant,
                                 uint32_t max_skips_per_pass /* 0: as the reference.  > 0: a ray that has stepped over this many empty voxels in one call stops with the
                                    samples it has (payload.n_steps < n_steps) and payload.alive = NGP_RAY_PAUSED; ngp_hip_composite composites them and keeps the ray
                                    alive, and the next call resumes the march where it stopped — the ray's samples, and so its pixel, are those of the reference's
                                    schedule, but a pass no longer lasts as long as its longest walk through empty space.  (Cost mode counts steps per pass: use 0.) */,
                                 const uint32_t* brick_summary /* as ngp_hip_advance_pos */,
                                 uint32_t* zero_word /* NULL, or a device word the kernel sets to 0 (the caller's next compaction counter) */); /* :705 */
/* render_mode = ERenderMode (common.h:80-91): AO 0, Shade 1, Normals 2 (network_input.pos holds d(density output)/d(pos), written there by
 * ngp_hip_nerf_input_gradient: the colour is normalize(-density'(out[3]) * that), :941-946), Positions 3 (show_accel >= 0: the occupancy-cell colouring),
 * Depth 4 (depth_scale = 1 / dataset scale, :2415), Distortion 5 (painted by init_rays), Cost 6 (n_steps / 128, in shade), Slice 7 (shade only),
 * EncodingVis 8 (network_input.pos holds what ngp_hip_nerf_visualize_activation wrote, :961-962).  The plain Shade frame is
 * (render_mode 1, depth_scale 1, show_accel -1, extras NULL). */
int ngp_hip_composite(void* stream, uint32_t n_elements, uint32_t current_step, const NgpAabb* aabb_host, const float* camera_matrix_host,
                      float* rgba, float* depth, NgpPayload* payloads, const NgpCoord* network_input, const uint16_t* network_output,
                      uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation, float min_transmittance,
                      int render_mode, float depth_scale, int show_accel, const NgpRenderExtras* extras_host, const NgpCompactOut* compact_host);   /* :767 */
int ngp_hip_shade(void* stream, uint32_t n_elements, const float* rgba, const float* depth, const NgpPayload* payloads,
                  int train_in_linear_colors, float* frame_buffer, float* depth_buffer, int render_mode);                      /* :1748 */
int ngp_hip_accumulate(void* stream, const int32_t* res_host, const float* frame_buffer, float* accumulate_buffer, float sample_count, int color_space); /* render_buffer.cu:235 */
int ngp_hip_tonemap(void* stream, const int32_t* res_host, float exposure, const float* background_color_host, const float* accumulate_buffer,
                    int color_space, int output_color_space, int tonemap_curve, int clamp_output_color, float* surface);       /* render_buffer.cu:540 */

/* [tcnn] GridEncodingTemplated::backward_impl -> kernel_grid_backward (tiny-cuda-nn encodings/grid.h; the reference reaches it through
 * m_network->backward, src/testbed_nerf.cu:3331): grid_grad[entry][f] = sum over samples and corners of half(w * dL/dx[level][f]),
 * EGradientMode::Overwrite.  dL_dx_planes: fp16 [16 levels][n] x 2 features (level-major planes, the layout ngp_hip_nerf_backward produces
 * internally); grid_grad: fp16 [n_grid_entries][2], every entry written.  Levels that are dense or hashed with a power-of-two table of at most 2^20 entries
 * are summed EXACTLY (64-bit fixed point, one rounding to fp16, independent of the order of the adds); tcnn rounds
 * after every atomicAdd(half2), so results agree to fp16 accumulation noise, not bit for bit.  n must be a multiple of 256. */
uint64_t ngp_hip_grid_backward_scratch_bytes(uint32_t n);
int ngp_hip_grid_backward(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const float* pos, uint32_t pos_stride_floats, uint32_t n,
                          const uint16_t* dL_dx_planes, uint16_t* grid_grad, void* scratch, uint64_t scratch_bytes);
/* the same sums for a batch that is NOT in ray order (image fitting, SDF: stratified / random positions): dense levels are binned as pair records like hashed ones instead of by the
 * run-merging walk, which finds nothing to merge there (what ngp_hip_gridmlp_backward runs).  Bit-identical results: the sums are exact either way. */
int ngp_hip_grid_backward_unordered(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const float* pos, uint32_t pos_stride_floats, uint32_t n,
                          const uint16_t* dL_dx_planes, uint16_t* grid_grad, void* scratch, uint64_t scratch_bytes);

/* ============================ plumbing configs P1 (2-D image) / P2 (SDF): grid encoding -> one FullyFusedMLP ============================
 * tcnn NetworkWithInputEncoding as Testbed::reset_network builds it for Image / Sdf mode (src/testbed.cu:2397-2445): HashGrid (16 levels x 2
 * features over n_dims = 2 or 3) -> 64 -> 64 -> 16 (ReLU, no output activation).  Parameters: [input 64x32 | hidden 64x64 | output 16x64 | grid].
 * Training step = forward, ngp_hip_loss_and_gradient, backward, ngp_hip_optimizer_step(n_params, 7168, ...)  (Trainer::training_step +
 * optimizer_step(128): src/testbed_image.cu:277-288, src/testbed_sdf.cu:1229-1252). */
int ngp_hip_gridmlp_make_desc_host(uint32_t n_dims, uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale, NgpNetDesc* out);
uint32_t ngp_hip_gridmlp_n_params_host(const NgpNetDesc* desc_host);
int ngp_hip_gridmlp_init_params(void* stream, const NgpNetDesc* desc_host, uint64_t seed, float* master, uint16_t* params, uint16_t* inference_params);
/* out: fp16, sample i at out[i*out_stride + 0..3] = network outputs 0..3; x_saved: NULL (inference) or [n][32] fp16 for backward */
int ngp_hip_gridmlp_forward(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const uint16_t* params, const float* pos, uint32_t pos_stride_floats,
                            uint32_t n, uint16_t* out, uint32_t out_stride, uint16_t* x_saved);
/* The same pass as two kernels (as ngp_hip_nerf_forward_ws above: XCD-affine encode of the 16 levels into planes inside `workspace`, then the MLP kernel): same
 * bits, faster for batches whose positions have no order (the SDF config: 2^18 random points), slower for batches that run along a dense level's x axis (the
 * stratified image batch).  The reference has one path (tcnn NetworkWithInputEncoding::forward behind Trainer::training_step, src/testbed_sdf.cu:1229-1252,
 * src/testbed_image.cu:277-288); which of the two runs is the host's measured choice (Testbed::network_pass).  workspace: ngp_hip_nerf_encode_workspace_bytes(n). */
int ngp_hip_gridmlp_forward_ws(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const uint16_t* params, const float* pos, uint32_t pos_stride_floats,
                               uint32_t n, uint16_t* out, uint32_t out_stride, uint16_t* x_saved, void* workspace, uint64_t workspace_bytes);
uint64_t ngp_hip_gridmlp_backward_scratch_bytes(uint32_t n);
/* dL_dout: fp16 [n][dl_stride], channels 0..3 consumed; grads: fp16 [n_params], overwritten.  n must be a multiple of 256. */
int ngp_hip_gridmlp_backward(void* stream, uint32_t n_dims, const NgpNetDesc* desc_dev, const uint16_t* params, const float* pos, uint32_t pos_stride_floats,
                             uint32_t n, const uint16_t* x_saved, const uint16_t* dL_dout, uint32_t dl_stride, uint16_t* grads, void* scratch, uint64_t scratch_bytes);
/* [tcnn] L2 / RelativeL2 / L1 / MAPE losses as Trainer::training_step applies them: predictions fp16 [n][pred_stride] (first `dims` channels), targets
 * fp32 [n][dims]; values fp32 [n][dims] (each already divided by n*dims: their sum is Trainer::loss); gradients fp16 [n][grad_stride], channels 0..3 written. */
int ngp_hip_loss_and_gradient(void* stream, int loss_type, uint32_t n, uint32_t dims, float loss_scale, const uint16_t* predictions, uint32_t pred_stride,
                              const float* targets, float* values, uint16_t* gradients, uint32_t grad_stride);
/* [tcnn] generate_random_uniform (call site src/testbed_image.cu:236): out[k] = k-th next_float() of pcg32{state, inc}; the host then advances by n_elements */
int ngp_hip_generate_random_uniform(void* stream, uint64_t rng_state, uint64_t rng_inc, uint32_t n_elements, float* out);
int ngp_hip_image_stratify2(void* stream, uint32_t n_elements, uint32_t log2_batch_size, float* inout_xy);                         /* src/testbed_image.cu:62-77 */
/* eval_image_kernel_and_snap<T, stride> (src/testbed_image.cu:172-218): texture = H*W texels of 4 x T, image_data_type 2 = half, 3 = float */
int ngp_hip_image_eval_and_snap(void* stream, uint32_t n_elements, const void* texture, int image_data_type, float* positions_xy, const int32_t* resolution_host,
                                float* result, uint32_t stride, int snap_to_pixel_centers, int linear_colors);
int ngp_hip_image_init_coords(void* stream, float* positions_xy, const int32_t* res_host, const int32_t* image_res_host, float view_dist, const float* image_pos_host,
                              const float* screen_center_host, int snap_to_pixel_centers, uint32_t sample_index);                                 /* src/testbed_image.cu:79-108 */
/* colours / predictions are the network's fp16 outputs ([n][stride], channels 0..2) instead of the reference's float3 matrices */
int ngp_hip_image_shade(void* stream, const int32_t* res_host, const float* positions_xy, const uint16_t* colors, uint32_t color_stride, float* frame_buffer,
                        float* depth_buffer, int linear_colors);                                                                               /* :139-170 */
int ngp_hip_image_coords_from_idx(void* stream, uint32_t n_elements, uint32_t offset, float* positions_xy, const int32_t* res_host);              /* :436-446 */
int ngp_hip_image_mse(void* stream, uint32_t n_elements, const float* target, const uint16_t* prediction, uint32_t pred_stride, float* result, int quantize_to_byte); /* :448-460 */

/* ============================ Blender multi-NeRF renderer (src/nerf_renderer.cu:17-563, include/.../nerf/ headers) ============================
 * One GLOBAL ray per output pixel (world space) and one PROXY ray per (pixel, NeRF) in that NeRF's local frame; per pass the nearest alive
 * proxy of every pixel is marched, evaluated by its own network and composited into the global ray.  4x4 matrices are column-major. */
typedef struct {                                                     /* render_data_workspace.cuh:13-20 NerfGlobalRay */
	float origin[3]; float dir[3]; float rgba[4]; uint32_t idx; float depth; uint8_t alive; uint8_t pad_[3];
} NgpGlobalRay;                                                      /* 52 B */
typedef struct {                                                     /* render_data_workspace.cuh:22-31 NerfProxyRay */
	float origin[3]; float dir[3]; float t; uint32_t idx; uint16_t n_steps; uint8_t alive; uint8_t active; float mask_alpha;
} NgpProxyRay;                                                       /* 40 B */
typedef struct {                                                     /* nerf_props.cuh:14-48 NerfProps */
	float transform[16], itransform[16];
	const uint8_t* density_grid_bitfield;   /* device, 8 cascades, max-pooled (may be NULL) */
	uint32_t grid_size, grid_volume;        /* 128, 128^3 */
	NgpAabb render_aabb, train_aabb;
	const NgpMask3D* masks;                 /* device */
	uint32_t n_masks;
	float cone_angle, min_cone_stepsize, max_cone_stepsize;
	uint32_t nerf_cascades;
	float opacity;
} NgpNerfProps;
typedef struct { int32_t max_res[2], scaled_res[2], skip[2]; uint32_t max_pixels, scaled_pixels; } NgpDownsampleInfo;   /* common.h:337-355 MakeFromMip */

int ngp_hip_multi_init_global_rays(void* stream, uint32_t sample_index, NgpGlobalRay* rays, float* depthbuffer, const NgpDownsampleInfo* ds_host,
                                   const NgpRenderCamera* camera_host);                                                        /* nerf_renderer.cu:17-94 */
int ngp_hip_multi_init_proxy_rays(void* stream, uint32_t n_elements, const NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays,
                                  const NgpNerfProps* nerf_props_dev);                                                         /* :96-146 */
int ngp_hip_multi_compact_rays(void* stream, uint32_t n_elements, const NgpGlobalRay* global_src, NgpGlobalRay* global_dst, const NgpProxyRay* proxy_src,
                               NgpProxyRay* proxy_dst, uint32_t n_nerfs, uint32_t proxy_stride_between_nerfs, NgpGlobalRay* global_final,
                               uint32_t* alive_counter, uint32_t* final_counter);                                              /* :237-268 */
int ngp_hip_multi_march_active_rays(void* stream, uint32_t n_rays_alive, uint32_t n_nerfs, const NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays,
                                    uint32_t proxy_stride_between_nerfs, const NgpNerfProps* nerf_props_dev);                  /* :272-316 (hit_test_and_march :148-208) */
int ngp_hip_multi_cull_rays(void* stream, uint32_t n_rays_alive, uint32_t n_nerfs, NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays,
                            uint32_t proxy_stride_between_nerfs, const float* cam_pos_host, const NgpNerfProps* nerf_props_dev); /* :378-428 */
int ngp_hip_multi_generate_next_inputs(void* stream, uint32_t n_elements, const NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays_of_nerf,
                                       NgpCoord* network_input, uint32_t n_steps, const NgpNerfProps* nerf_props_of_nerf_dev);  /* :318-375 */
/* network_output is fp16 [n][out_stride] (r, g, b, sigma first), i.e. the layout ngp_hip_nerf_inference writes */
int ngp_hip_multi_composite(void* stream, uint32_t n_global_rays, uint32_t current_step, NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays_of_nerf,
                            const NgpCoord* network_input, const uint16_t* network_output, uint32_t out_stride, uint32_t n_steps, int rgb_activation,
                            int density_activation, float min_transmittance, const NgpNerfProps* nerf_props_of_nerf_dev);      /* :431-510 */
/* Sparse variants (not in the reference, which evaluates every alive ray's slots for every NeRF): ngp_hip_multi_cull_rays_collect also
 * appends ray i to active_lists[nerf * stride + ...] of the NeRF whose proxy ray it activated (active_counts[n_nerfs], zeroed inside);
 * the _list kernels then run thread k on ray list[k] with a network batch of exactly those n rays (element k + j * n).  Same pixels. */
int ngp_hip_multi_cull_rays_collect(void* stream, uint32_t n_rays_alive, uint32_t n_nerfs, NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays, uint32_t stride, const float* cam_pos,
                                    const NgpNerfProps* props, uint32_t* active_lists, uint32_t* active_counts);
int ngp_hip_multi_generate_next_inputs_list(void* stream, uint32_t n_elements, const uint32_t* list, const NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays, NgpCoord* network_input,
                                            uint32_t n_steps, const NgpNerfProps* props);
int ngp_hip_multi_composite_list(void* stream, uint32_t n_global_rays, const uint32_t* list, uint32_t current_step, NgpGlobalRay* global_rays, NgpProxyRay* proxy_rays,
                                 const NgpCoord* network_input, const uint16_t* network_output, uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation,
                                 float min_transmittance, const NgpNerfProps* props);
/* One launch at the head of a pass = march_active_rays + cull_global_rays_and_set_proxy_rays_active + compact_rays (nerf_renderer.cu:675-733, in that order: a ray
 * whose last proxy dies leaves one pass earlier than in the reference, where it idles through that pass; same colours) + the per-NeRF lists of
 * ngp_hip_multi_cull_rays_collect (indices into the COMPACTED arrays) + the pass's counts.  proxy_src is updated in place (t, alive, active) before it is copied.
 * max_skips > 0: a proxy that has walked that many empty voxels keeps its t and the ray sits this pass out (no cull, no samples; the walk resumes next pass at the
 * same DDA iteration).  counters: [0] alive, [1 + n] rays that sample NeRF n — zero at launch; next_counters: the set the launch's last workgroup clears for the next
 * pass; host_mailbox (optional, device pointer to 1 + n_nerfs mapped host words): word k = counters[k] | sequence << 32, written by the last workgroup.
 * tile_w, tile_h > 0: the threads visit the rays in 8 x 8 pixel tiles of a tile_w x tile_h image (ray i = pixel (i % tile_w, i / tile_w)): the first pass. */
int ngp_hip_multi_advance(void* stream, uint32_t n_prev, uint32_t n_nerfs, const NgpGlobalRay* global_src, NgpProxyRay* proxy_src, NgpGlobalRay* global_dst, NgpProxyRay* proxy_dst,
                          uint32_t stride, NgpGlobalRay* global_final, const float* cam_pos, const NgpNerfProps* props, uint32_t max_skips, uint32_t* counters, uint32_t* next_counters,
                          uint32_t* final_counter, uint32_t* active_lists, uint32_t* blocks_done, uint64_t* host_mailbox, uint32_t sequence, uint32_t tile_w, uint32_t tile_h);
int ngp_hip_multi_shade(void* stream, uint32_t n_rays, const NgpGlobalRay* rays, int train_in_linear_colors, float* frame_buffer, float* depth_buffer,
                        const NgpDownsampleInfo* ds_host, int flip_y);                                                        /* :512-563 */

/* ============================ stock renderer, the rest of row f3: Normals / EncodingVis / Slice passes ============================ */
/* [tcnn] DifferentiableObject::input_gradient(stream, dim, input, d_output_d_input = input) as the tracer calls it for the Normals mode
 * (src/testbed_nerf.cu:2225-2226; also :3432 for the marching-cubes normals): dL/doutput = backprop_scale (128) at output `dim` and 0 elsewhere,
 * forward + backward of NerfNetwork (nerf_network.h:143-266) without parameter gradients, then every element of the input matrix times
 * 1 / backprop_scale — the position and direction rows then hold d out[dim] / d input, the dt row its old value / 128 (no gradient is written
 * to it).  coords: [n] NgpCoord, overwritten in place.  n must be a multiple of 256.  scratch: ngp_hip_nerf_input_gradient_scratch_bytes(n). */
uint64_t ngp_hip_nerf_input_gradient_scratch_bytes(uint32_t n);
int ngp_hip_nerf_input_gradient(void* stream, const NgpNetDesc* desc_dev, const NgpNetDesc* desc_host, const uint16_t* params, uint32_t dim, float* coords_inout,
                                uint32_t coord_stride_floats, uint32_t n, void* scratch, uint64_t scratch_bytes, const NgpNetVariant* variant /* NULL: base family */);
/* [tcnn] Network::visualize_activation(stream, layer, dimension, input, output): forward pass, then extract_dimension_pos_neg_kernel on
 * forward_activations(layer) — NerfNetwork's layer map (nerf_network.h:474-501): 0 = hash encoding (32), 1 = density hidden layer (64),
 * 2 = colour network input [density output 16 | SH 16], 3.. = colour hidden layers (64).  Output row 0 = max(-v, 0), row 1 = max(v, 0),
 * row 2 = 0, every further row = 1; out_stride_floats = 7 writes over the NgpCoord it read (the tracer's EncodingVis, src/testbed_nerf.cu:2227-2228),
 * 4 gives the rgba of the Slice mode (:2462). */
int ngp_hip_nerf_visualize_activation(void* stream, const NgpNetDesc* desc_dev, const uint16_t* params, uint32_t layer, uint32_t dimension, const float* coords,
                                      uint32_t coord_stride_floats, uint32_t n, float* out, uint32_t out_stride_floats, const NgpNetVariant* variant /* NULL: base family */);
/* Slice mode (src/testbed_nerf.cu:2445-2476): network inputs at payload.origin + dir * t (generate_nerf_network_inputs_at_current_position, :676-682),
 * fp16 network outputs -> fp32 rgba with alpha = clamp(1 - exp(-density * depth), 0, 1) and colours premultiplied (compute_nerf_rgba, :684-703). */
int ngp_hip_generate_inputs_at_current_position(void* stream, uint32_t n_elements, const NgpAabb* aabb_host, const NgpPayload* payloads, NgpCoord* network_input);
int ngp_hip_compute_nerf_rgba(void* stream, uint32_t n_elements, const uint16_t* network_output, uint32_t out_stride, float* rgba, int rgb_activation, int density_activation,
                              float depth, int density_as_alpha);

/* ---- data-parallel collectives: one RCCL communicator per rank (one process per GPU; xGMI inside the node).  The reference is
 * single-GPU (README.md:239-241); these are the calls its training step would make between backward and optimizer_step (src/testbed_nerf.cu:3331,
 * 2950) and around NerfCounters::update_after_training (2870-2894).  RCCL is bound at run time: a copy the process already carries (PyTorch
 * bundles one with its own HIP runtime) is used, else /opt/rocm/lib/librccl.so.1.  All-reduces are sums, in place, ordered on `stream`. */
int ngp_rccl_available(void);                                  /* 1 if an RCCL library could be bound */
int ngp_rccl_get_unique_id(uint8_t* out128);                   /* ncclGetUniqueId: rank 0 makes one, every rank passes the same 128 bytes to init */
void* ngp_rccl_init(int rank, int world_size, const uint8_t* unique_id128);   /* ncclCommInitRank (collective; the rank's device is current); NULL on failure */
int ngp_rccl_allreduce_grads(void* comm, void* stream, uint16_t* grads_f16, uint64_t n_params);   /* fp16 sum of the loss-scaled gradient vector */
int ngp_rccl_allreduce_f32(void* comm, void* stream, float* values, uint64_t count);              /* error maps, exposure gradients */
int ngp_rccl_allreduce_counters(void* comm, void* stream, double* values, uint64_t count);        /* {samples, compacted samples, loss sum} of a step */
/* in place: rank r's chunk sits at buf + r * count_per_rank elements before the call, every rank holds all world chunks after it
 * (frame rows of a sharded render; the parameter shards behind the sharded optimizer step) */
int ngp_rccl_allgather_f32(void* comm, void* stream, float* buf, uint64_t count_per_rank);
int ngp_rccl_allgather_f16(void* comm, void* stream, uint16_t* buf, uint64_t count_per_rank);
/* sum over the ranks of in[world * count_per_rank]; rank r receives elements [r * count_per_rank, (r + 1) * count_per_rank) of the sum (fp32: the
 * gradient shards of the sharded optimizer step are summed in fp32, not in the fp16 they are stored in) */
int ngp_rccl_reduce_scatter_f32(void* comm, void* stream, const float* in, float* out, uint64_t count_per_rank);
/* The gradient exchange with fp16 on the wire (round 5; replaces widen -> ngp_rccl_reduce_scatter_f32 -> narrow in the sharded optimizer step): slice q of `send`
 * (world x count_per_rank fp16) goes to rank q, rank q's slice for this rank arrives in slice q of `recv` — grouped ncclSend / ncclRecv over the node's direct xGMI
 * links, no arithmetic on the wire, the own slice copied on the device.  ngp_hip_sum_slices_f16 then gives out[i] = half(sum_q float(slices[q][i])) with the ranks
 * added IN RANK ORDER in fp32 and ONE fp16 rounding: the gradient a replicated step would see, independent of the collective library's reduction order, at
 * (world - 1) / world x 2 bytes per parameter and rank instead of 4.  The reference has no counterpart (single GPU, README.md:239-241); call site: between
 * backward (src/testbed_nerf.cu:3331) and optimizer_step (2950). */
int ngp_rccl_alltoall_f16(void* comm, void* stream, const uint16_t* send, uint16_t* recv, uint64_t count_per_rank);
int ngp_hip_sum_slices_f16(void* stream, uint32_t world, uint32_t count, const uint16_t* slices, uint16_t* out);
int ngp_rccl_comm_size(void* comm);                              /* ncclCommCount; -1 on error */
int ngp_rccl_comm_rank(void* comm);
int ngp_rccl_finalize(void* comm);

/* ============================ measuring stick (not a stage of the path) ============================
 * Random 4-byte gathers from table[0 .. n_entries) — n_blocks x 256 threads, per_thread independent loads each; per_xcd != 0 confines every XCD to its own eighth of
 * the table.  A benchmark times launches of it (HIP events) to state the request-rate roofs of the box it runs on — vector-L1 reach, one XCD's L2, the Infinity Cache
 * across the fabric — next to the byte roof (bench.py `roofline.request_rate`).  *n_gathers_out (host): gathers per launch. */
int ngp_hip_probe_gather_rate(void* stream, const uint32_t* table, uint32_t n_entries, int per_xcd, uint32_t n_blocks, uint32_t per_thread, uint32_t seed, uint32_t* sink, uint64_t* n_gathers_out);

#ifdef __cplusplus
}
#endif
#endif
