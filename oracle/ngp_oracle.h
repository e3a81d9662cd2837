/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.  See orc_core.h for the full header note.
 * Public (ctypes-visible) surface of liborc: every function cites its reference file:line at the definition.
 */
#ifndef NGP_ORACLE_H
#define NGP_ORACLE_H

#include "orc_core.h"

#ifdef __cplusplus
extern "C" {
#endif

/* common.h:103-111 ELossType */
enum { ORC_LOSS_L2 = 0, ORC_LOSS_L1 = 1, ORC_LOSS_MAPE = 2, ORC_LOSS_SMAPE = 3, ORC_LOSS_HUBER = 4, ORC_LOSS_LOG_L1 = 5, ORC_LOSS_RELATIVE_L2 = 6 };

typedef struct { float scale; uint32_t resolution; uint32_t offset; uint32_t size; } orc_grid_level;
typedef struct {
	uint32_t n_levels;          /* 16 */
	uint32_t n_grid_entries;    /* sum of level sizes (x2 features = grid params) */
	orc_grid_level levels[16];
} orc_net;
typedef struct { uint16_t x[32], h1[64], in_rgb[32], h2[64], h3[64], out[16]; } orc_act;

/* orc_sampling.c */
float orc_ld_random_val_export(uint32_t index, uint32_t seed, uint32_t dim);
void orc_ld_random_pixel_offset_export(uint32_t spp, float* out);
uint32_t orc_morton3D_export(uint32_t x, uint32_t y, uint32_t z);
uint32_t orc_morton3D_invert_export(uint32_t x);
uint16_t orc_f2h_export(float f);
float orc_h2f_export(uint16_t h);
void orc_pcg32_floats(uint64_t seed, int64_t advance, uint32_t n, float* out, uint64_t* state_out);
void orc_pcg32_uints(uint64_t seed, int64_t advance, uint32_t n, uint32_t* out);
void orc_mark_untrained_density_grid(uint32_t n_elements, float* grid_out, uint32_t n_training_images, const orc_image_meta* metadata, const orc_xform* xforms, int clear_visible_voxels);
void orc_generate_grid_samples_nonuniform(uint32_t n_elements, uint64_t rng_state, uint64_t rng_inc, uint32_t step, const orc_aabb* aabb, const float* grid_in, float* out_pos, uint32_t* indices, uint32_t n_cascades, float thresh);
void orc_splat_grid_samples_max(uint32_t n_elements, const uint32_t* indices, const uint16_t* network_output, float* grid_out, int density_activation);
void orc_ema_grid_samples(uint32_t n_elements, float decay, float* grid_out, const float* grid_in);
float orc_density_grid_mean(const float* grid, uint32_t n_elements);
void orc_grid_to_bitfield(uint32_t n_elements, uint32_t n_nonzero_elements, const float* grid, uint8_t* bitfield, float mean_density);
void orc_bitfield_max_pool(uint32_t n_elements, const uint8_t* prev_level, uint8_t* next_level);
void orc_update_bitfield(const float* grid, uint32_t n_cascades_used, float mean_density, uint8_t* bitfield);
void orc_iterative_opencv_lens_undistortion(const float* params, float* u, float* v);
void orc_get_xform_given_rolling_shutter(const orc_xform* xf, const float rs[4], float u, float v, float motionblur_time, float out[12]);
void orc_read_rgba(const float xy[2], const int32_t res[2], const void* pixels, int type, float out[4]);
void orc_nerf_random_image_pos_training(orc_pcg32* rng, const int32_t res[2], int snap_to_pixel_centers, const orc_error_map_cdf* cdf, uint32_t img, float xy[2], float* pdf);
uint32_t orc_image_idx(uint32_t base_idx, uint32_t n_rays, uint32_t n_training_images, const float* cdf, float* pdf);
uint32_t orc_binary_search(float val, const float* data, uint32_t length);
void orc_construct_cdf_2d(uint32_t n_images, uint32_t height, uint32_t width, const float* data, float* cdf_x_cond_y, float* cdf_y);
void orc_construct_cdf_1d(uint32_t n_images, uint32_t height, float* cdf_y, float* cdf_img);
void orc_image_cdf_host(uint32_t n_images, const float* pmf_unnormalized, float* pmf_out, float* cdf_out);
void orc_generate_training_samples(
	uint32_t n_rays, const orc_aabb* aabb, uint32_t max_samples, uint64_t rng_state, uint64_t rng_inc,
	uint32_t* ray_counter, uint32_t* numsteps_counter, uint32_t* ray_indices_out, orc_ray* rays_out_unnormalized,
	uint32_t* numsteps_out, orc_coord* coords_out, uint32_t n_training_images, const orc_image_meta* metadata,
	const orc_xform* xforms, const uint8_t* density_grid, int max_level_rand_training, float* max_level_ptr,
	int snap_to_pixel_centers, int train_envmap, float cone_angle_constant, const float* distortion_data,
	const int32_t distortion_resolution[2], uint32_t ray_offset, uint32_t n_rays_global, const orc_error_map_cdf* cdf /* NULL: uniform */);

/* orc_network.c */
void orc_net_make_levels(uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale, orc_grid_level* levels, uint32_t* n_grid_entries);
float orc_per_level_scale(uint32_t n_levels, uint32_t base_resolution, float desired_resolution, uint32_t aabb_scale);
uint32_t orc_net_n_params(const orc_net* net);
uint32_t orc_net_mlp_params(void);
void orc_grid_encode_one(const orc_net* net, const uint16_t* grid, const float pos_in[3], uint16_t* out);
void orc_sh4(const float dir[3], float out[16]);
void orc_nerf_forward_one(const orc_net* net, const uint16_t* params, const float coord[7], orc_act* a);
void orc_nerf_inference(const orc_net* net, const uint16_t* params, const float* coords, uint32_t coord_stride_floats, uint32_t n, uint16_t* out, uint32_t out_stride);
void orc_nerf_density(const orc_net* net, const uint16_t* params, const float* pos, uint32_t pos_stride_floats, uint32_t n, uint16_t* out0);
void orc_nerf_forward_backward(const orc_net* net, const uint16_t* params, const float* coords, uint32_t coord_stride_floats, uint32_t n, const uint16_t* dL_dout, uint16_t* out_rgbsigma, double* grads_out, uint16_t* dL_dx_out);
void orc_nerf_input_gradient(const orc_net* net, const uint16_t* params, const float* coords, uint32_t coord_stride_floats, uint32_t n, const uint16_t* dL_dout, float* dL_dinput /* [n][6] */);
void orc_nerf_visualize_activation(const orc_net* net, const uint16_t* params, uint32_t layer, uint32_t dimension, const float* coords, uint32_t coord_stride_floats, uint32_t n, float* out, uint32_t out_stride_floats);
void orc_nerf_init_params(const orc_net* net, uint64_t seed, float* params_fp32);
uint32_t orc_grid_index_export(const orc_grid_level* lv, uint32_t x, uint32_t y, uint32_t z);
/* orc_netx.c — NerfNetwork for configs other than configs/nerf/base.json: per-image extra dims (latent codes / light directions) behind the direction encoding, and
 * 0..3 hidden layers in the colour network.  extra_dims: [rows][n_extra_dims] fp32; sample_slot: per sample the row to use (NULL: row 0 for every sample) */
typedef struct { uint32_t n_extra_dims, n_rgb_hidden_layers; const float* extra_dims; const uint32_t* sample_slot; } orc_netx;
uint32_t orc_netx_mlp_params(const orc_netx* x);
uint32_t orc_netx_n_params(const orc_net* net, const orc_netx* x);
void orc_nerf_inference_x(const orc_net* net, const orc_netx* x, const uint16_t* params, const float* coords, uint32_t coord_stride_floats, uint32_t n, uint16_t* out, uint32_t out_stride);
void orc_nerf_forward_backward_x(const orc_net* net, const orc_netx* x, const uint16_t* params, const float* coords, uint32_t coord_stride_floats, uint32_t n, const uint16_t* dL_dout, uint16_t* out_rgbsigma, double* grads_out, uint16_t* dL_dx_out, float* dL_dextra_out);
void orc_nerf_init_params_x(const orc_net* net, const orc_netx* x, uint64_t seed, float* params_fp32);
void orc_compute_extra_dims_gradient(uint32_t n_rays_alive, const uint32_t* ray_image, const uint32_t* numsteps, const float* dL_dextra, uint32_t n_extra_dims, float* gradient);
void orc_f32_to_f16(const float* in, uint16_t* out, uint32_t n);
void orc_f16_to_f32(const uint16_t* in, float* out, uint32_t n);
void orc_adam_ema_step(uint32_t n_params, uint32_t n_matrix_params, uint32_t step, float base_lr_after_decay, float beta1, float beta2, float epsilon, float l2_reg, float loss_scale, float ema_decay, const uint16_t* grads_fp16, float* master, uint16_t* params_fp16, float* m1, float* m2, float* ema_fp32, uint16_t* inference_fp16);

/* orc_network.c — plumbing configs P1 / P2 (grid encoding -> one MLP) */
void orc_gridmlp_make_levels(uint32_t n_dims, uint32_t n_levels, uint32_t log2_hashmap_size, uint32_t base_resolution, float per_level_scale, orc_grid_level* levels, uint32_t* n_grid_entries);
uint32_t orc_gridmlp_n_params(const orc_net* net);
void orc_grid_encode_nd(uint32_t n_dims, const orc_net* net, const uint16_t* grid, const float* pos_in, uint16_t* out);
void orc_gridmlp_inference(uint32_t n_dims, const orc_net* net, const uint16_t* params, const float* pos, uint32_t pos_stride_floats, uint32_t n, uint16_t* out, uint32_t out_stride);
/* [tcnn] kernel_grid_backward with the sum taken exactly: out[entry][f] = fp16( sum of half(w * dL/dx) ), ONE rounding.  dL_dx_planes fp16 [n_levels][n][2]. */
void orc_grid_backward_exact(uint32_t n_dims, const orc_net* net, const float* pos_all, uint32_t pos_stride_floats, uint32_t n, const uint16_t* dL_dx_planes, uint16_t* grid_grad);
void orc_gridmlp_forward_backward(uint32_t n_dims, const orc_net* net, const uint16_t* params, const float* pos_all, uint32_t pos_stride_floats, uint32_t n, const uint16_t* dL_dout, uint16_t* out4, double* grads_out, uint16_t* dL_dx_out);
void orc_gridmlp_init_params(const orc_net* net, uint64_t seed, float* params_fp32);
void orc_tcnn_loss_and_gradient(int loss_type, uint32_t n, uint32_t dims, float loss_scale, const uint16_t* predictions, uint32_t pred_stride, const float* targets, float* values, uint16_t* gradients, uint32_t grad_stride);
void orc_generate_random_uniform(uint64_t rng_state, uint64_t rng_inc, uint32_t n_elements, float* out);
void orc_image_stratify2(uint32_t n_elements, uint32_t log2_batch_size, float* inout_xy);
void orc_image_eval_and_snap(uint32_t n_elements, const void* texture, int image_data_type, float* positions_xy, const int32_t res[2], float* result, uint32_t stride, int snap_to_pixel_centers, int linear_colors);

/* orc_loss.c */
void orc_loss_and_gradient_export(const float* target, const float* prediction, int loss_type, float* loss, float* grad);
void orc_compute_loss(
	uint32_t n_rays, const orc_aabb* aabb, uint64_t rng_state, uint64_t rng_inc, uint32_t max_samples_compacted, uint32_t n_rays_alive,
	float loss_scale, uint32_t mlp_stride, const float background_color_in[3], int color_space_srgb, int train_with_random_bg_color,
	int train_in_linear_colors, uint32_t n_training_images, const orc_image_meta* metadata, const uint16_t* network_output,
	uint32_t* numsteps_counter, const uint32_t* ray_indices_in, const orc_ray* rays_in_unnormalized, uint32_t* numsteps_in,
	const orc_coord* coords_in_all, orc_coord* coords_out_all, uint16_t* dloss_doutput_all, int loss_type, float* loss_output,
	int max_level_rand_training, float* max_level_compacted_ptr_all, int rgb_activation, int density_activation, int snap_to_pixel_centers,
	float* error_map, const int32_t error_map_res[2], float mean_density, const float* exposure, float near_distance, const orc_error_map_cdf* cdf /* NULL: uniform */,
	const uint16_t* encoded_in, uint16_t* encoded_out /* optional [sample][32] fp16 rows carried through the compaction */,
	float depth_supervision_lambda, int depth_loss_type, float* exposure_gradient /* NULL or [n_images][3], accumulated */);
void orc_compute_cam_gradient(
	uint32_t n_rays, const orc_aabb* aabb, uint64_t rng_state, uint64_t rng_inc, uint32_t n_rays_alive, int snap_to_pixel_centers,
	float* cam_pos_gradient, float* cam_rot_gradient, uint32_t n_training_images, const orc_image_meta* metadata, const uint32_t* ray_indices_in,
	const orc_ray* rays_in_unnormalized, const uint32_t* numsteps_in, const orc_coord* coords_all, const float* coords_gradient_all /* [sample][6] */,
	const orc_error_map_cdf* cdf);
typedef struct { const float* envmap_data; float* envmap_gradient; int32_t envmap_res[2]; int32_t envmap_loss_type;
                 const float* sharpness_data; int32_t sharpness_res[2]; float* sharpness_grid; } orc_loss_extras;   /* NgpLossExtras */
void orc_compute_sharpness(const int32_t sharpness_res[2], const int32_t image_res[2], const void* pixels, int image_data_type, float* sharpness_out);
void orc_compute_loss_ex(
	uint32_t n_rays, const orc_aabb* aabb, uint64_t rng_state, uint64_t rng_inc, uint32_t max_samples_compacted, uint32_t n_rays_alive, float loss_scale, uint32_t mlp_stride,
	const float background_color_in[3], int color_space_srgb, int train_with_random_bg_color, int train_in_linear_colors, uint32_t n_training_images, const orc_image_meta* metadata,
	const uint16_t* network_output, uint32_t* numsteps_counter, const uint32_t* ray_indices_in, const orc_ray* rays_in_unnormalized, uint32_t* numsteps_in, const orc_coord* coords_in_all,
	orc_coord* coords_out_all, uint16_t* dloss_doutput_all, int loss_type, float* loss_output, int max_level_rand_training, float* max_level_compacted_ptr_all, int rgb_activation,
	int density_activation, int snap_to_pixel_centers, float* error_map, const int32_t error_map_res[2], float mean_density, const float* exposure, float near_distance,
	const orc_error_map_cdf* cdf, const uint16_t* encoded_in, uint16_t* encoded_out, float depth_supervision_lambda, int depth_loss_type, float* exposure_gradient, const orc_loss_extras* ex);
void orc_compute_cam_gradient_ex(
	uint32_t n_rays, const orc_aabb* aabb, uint64_t rng_state, uint64_t rng_inc, uint32_t n_rays_alive, int snap_to_pixel_centers, float* cam_pos_gradient, float* cam_rot_gradient,
	uint32_t n_training_images, const orc_image_meta* metadata, const uint32_t* ray_indices_in, const orc_ray* rays_in_unnormalized, const uint32_t* numsteps_in, const orc_coord* coords_all,
	const float* coords_gradient_all, const orc_error_map_cdf* cdf, const orc_xform* xforms, float* distortion_gradient, float* distortion_gradient_weight, const int32_t* distortion_resolution);
void orc_safe_divide(uint32_t n, float* inout, const float* divisor);
void orc_optimizer_step_f32(uint32_t n, uint32_t step, float base_lr_after_decay, float beta1, float beta2, float epsilon, float loss_scale, float ema_decay, const float* grads, float* params,
                            float* m1, float* m2, float* ema);
void orc_image_from_rgba32_f16(uint64_t n_pixels, const uint8_t* rgba8, uint16_t* out_half4, uint32_t mask_color);
void orc_image_sharpen(uint64_t n_pixels, uint32_t w, const void* pix, void* dest, int is_half, float sharpen_amount);
void orc_fill_rollover_and_rescale_f16(uint32_t n_elements, uint32_t stride, uint32_t n_input_elements, uint16_t* inout);
void orc_fill_rollover_f32(uint32_t n_elements, uint32_t stride, uint32_t n_input_elements, float* inout);

/* orc_render.c */
void orc_extra_camera_model_pixel_to_ray(int model, uint32_t spp, uint32_t x, uint32_t y, float rx, float ry, const float* c, float sq_width, float sq_height, float sq_curvature,
                                         const float* qh_front, const float* qh_back, float near_distance, float focus_z, float aperture_size, orc_vec3* origin, orc_vec3* dir);
void orc_init_rays(uint32_t sample_index, orc_payload* payloads, const int32_t res[2], const float focal_length[2], const float* camera_matrix0, const float* camera_matrix1, const float rolling_shutter[4], const float screen_center[2], const float parallax_shift[3], int snap_to_pixel_centers, const orc_aabb* render_aabb, const float* render_aabb_to_local, float near_distance, int lens_mode, const float* lens_params, float* depthbuffer, float plane_z, float aperture_size, const orc_render_camera* camera_models);
void orc_advance_pos(uint32_t n_elements, const orc_aabb* render_aabb, const float* render_aabb_to_local, uint32_t sample_index, orc_payload* payloads, const uint8_t* density_grid, uint32_t min_mip, float cone_angle_constant);
void orc_compact_rays(uint32_t n_elements, const float* src_rgba, const float* src_depth, const orc_payload* src_payloads, float* dst_rgba, float* dst_depth, orc_payload* dst_payloads, float* dst_final_rgba, float* dst_final_depth, orc_payload* dst_final_payloads, uint32_t* counter, uint32_t* final_counter);
void orc_generate_next_inputs(uint32_t n_elements, const orc_aabb* render_aabb, const orc_aabb* train_aabb, orc_payload* payloads, orc_coord* network_input, uint32_t n_steps, const uint8_t* density_grid, uint32_t min_mip, float cone_angle_constant);
void orc_composite(uint32_t n_elements, uint32_t current_step, const orc_aabb* aabb, const float* camera_matrix, float* rgba, float* depth, orc_payload* payloads, const orc_coord* network_input, const uint16_t* network_output, uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation, float min_transmittance);
void orc_shade(uint32_t n_elements, const float* rgba, const float* depth, const orc_payload* payloads, int train_in_linear_colors, float* frame_buffer, float* depth_buffer);
void orc_composite_mode(uint32_t n_elements, uint32_t current_step, const orc_aabb* aabb, const float* camera_matrix, float* rgba, float* depth, orc_payload* payloads,
                        const orc_coord* network_input, const uint16_t* network_output, uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation,
                        float min_transmittance, int render_mode, float depth_scale, int show_accel);
void orc_shade_mode(uint32_t n_elements, const float* rgba, const float* depth, const orc_payload* payloads, int train_in_linear_colors, float* frame_buffer, float* depth_buffer,
                    int render_mode);
typedef struct {   /* NgpRenderExtras of include/ngp_hip.h */
	const orc_mask3d* render_masks; uint32_t n_render_masks; int32_t glow_mode; float glow_y_cutoff; const float* envmap; int32_t envmap_res[2];
	const float* distortion; int32_t distortion_res[2]; int32_t quilting_dims[2]; int32_t render_mode; float* frame_buffer;
	int32_t row_begin, row_end;   /* {0,0}: the whole frame; else init_rays sets up the rows [row_begin, row_end) only (a shard of the frame) */
	int32_t tile_order;           /* (the product's 8 x 8 tile slot order; the oracle keeps row-major slots) */
} orc_render_extras;
void orc_read_image2(const float* data, const int32_t res[2], const float pos[2], float out[2]);
void orc_read_envmap(const float* data, const int32_t res[2], const float dir[3], float out[4]);
void orc_init_rays_ex(uint32_t sample_index, orc_payload* payloads, const int32_t res[2], const float focal_length[2], const float* camera_matrix0, const float* camera_matrix1, const float rolling_shutter[4], const float screen_center[2], const float parallax_shift[3], int snap_to_pixel_centers, const orc_aabb* render_aabb, const float* render_aabb_to_local, float near_distance, int lens_mode, const float* lens_params, float* depthbuffer, float plane_z, float aperture_size, const orc_render_camera* camera_models, const orc_render_extras* ex);
void orc_composite_ex(uint32_t n_elements, uint32_t current_step, const orc_aabb* aabb, const float* camera_matrix, float* rgba, float* depth, orc_payload* payloads,
                      const orc_coord* network_input, const uint16_t* network_output, uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation,
                      float min_transmittance, int render_mode, float depth_scale, int show_accel, const orc_render_extras* ex);
void orc_generate_inputs_at_current_position(uint32_t n_elements, const orc_aabb* aabb, const orc_payload* payloads, orc_coord* network_input);
void orc_compute_nerf_rgba(uint32_t n_elements, const uint16_t* network_output, uint32_t out_stride, float* rgba, int rgb_activation, int density_activation, float depth, int density_as_alpha);
void orc_accumulate(const int32_t res[2], const float* frame_buffer, float* accumulate_buffer, float sample_count, int color_space_srgb);
void orc_tonemap(const int32_t res[2], float exposure, const float background_color_in[4], const float* accumulate_buffer, int color_space_srgb, int output_color_space_srgb, int tonemap_curve, int clamp_output_color, float* surface);
uint64_t orc_render_nerf(const orc_net* net, const uint16_t* inference_params, uint32_t sample_index, const int32_t res[2], const float focal_length[2], const float* camera_matrix0, const float* camera_matrix1, const float screen_center[2], int snap_to_pixel_centers, const orc_aabb* render_aabb, const float* render_aabb_to_local, const orc_aabb* train_aabb, float near_distance, const uint8_t* density_grid, float cone_angle_constant, int rgb_activation, int density_activation, float min_transmittance, int train_in_linear_colors, float* frame_buffer, float* depth_buffer);
uint64_t orc_render_nerf_rows(const orc_net* net, const uint16_t* inference_params, uint32_t sample_index, const int32_t res[2], const float focal_length[2], const float* camera_matrix0, const float* camera_matrix1, const float screen_center[2], int snap_to_pixel_centers, const orc_aabb* render_aabb, const float* render_aabb_to_local, const orc_aabb* train_aabb, float near_distance, const uint8_t* density_grid, float cone_angle_constant, int rgb_activation, int density_activation, float min_transmittance, int train_in_linear_colors, float* frame_buffer, float* depth_buffer, int row_begin, int row_end);

/* orc_multi.c — Blender multi-NeRF renderer (src/nerf_renderer.cu) */
float orc_mask_sample(const orc_mask3d* m, const float p[3]);
int orc_mask_intersects_ray(const orc_mask3d* m, const float ro[3], const float rd[3]);
void orc_downsample_info_from_mip(const int32_t resolution[2], uint32_t mip, orc_downsample_info* ds);
void orc_multi_init_global_rays(uint32_t sample_index, orc_global_ray* rays, float* depthbuffer, const orc_downsample_info* ds, const orc_render_camera* cam);
void orc_multi_init_proxy_rays(uint32_t n_elements, const orc_global_ray* global_rays, orc_proxy_ray* proxy_rays, const orc_nerf_props* props);
void orc_multi_compact_rays(uint32_t n_elements, const orc_global_ray* g_src, orc_global_ray* g_dst, const orc_proxy_ray* p_src, orc_proxy_ray* p_dst, uint32_t n_nerfs, uint32_t stride, orc_global_ray* g_final, uint32_t* alive_counter, uint32_t* final_counter);
void orc_multi_march_active_rays(uint32_t n_rays_alive, uint32_t n_nerfs, const orc_global_ray* global_rays, orc_proxy_ray* proxy_rays, uint32_t stride, const orc_nerf_props* props);
void orc_multi_cull_rays(uint32_t n_rays_alive, uint32_t n_nerfs, orc_global_ray* global_rays, orc_proxy_ray* proxy_rays, uint32_t stride, const float cam_pos[3], const orc_nerf_props* props);
void orc_multi_generate_next_inputs(uint32_t n_elements, const orc_global_ray* global_rays, orc_proxy_ray* proxy_rays, orc_coord* network_input, uint32_t n_steps, const orc_nerf_props* props);
void orc_multi_composite(uint32_t n_global_rays, uint32_t current_step, orc_global_ray* global_rays, orc_proxy_ray* proxy_rays, const orc_coord* network_input, const uint16_t* network_output, uint32_t out_stride, uint32_t n_steps, int rgb_activation, int density_activation, float min_transmittance, const orc_nerf_props* props);
void orc_multi_shade(uint32_t n_rays, const orc_global_ray* rays, int train_in_linear_colors, float* frame_buffer, float* depth_buffer, const orc_downsample_info* ds, int flip_y);
uint64_t orc_multi_render(uint32_t n_nerfs, const orc_net* const* nets, const uint16_t* const* params, const orc_nerf_props* props, const int* rgb_activation, const int* density_activation, const float* min_transmittance, const orc_downsample_info* ds, const orc_render_camera* cam, int flip_y, float* frame_buffer, float* depth_buffer);

#ifdef __cplusplus
}
#endif
#endif
