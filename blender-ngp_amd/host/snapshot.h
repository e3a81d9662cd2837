// snapshot.h — `.msgpack` snapshot schema shared by Testbed::save_snapshot / load_snapshot (src/testbed.cu:3006-3106) and the Blender
// renderer's per-NeRF loader (NeuralRadianceField::load_snapshot, include/neural-graphics-primitives/nerf/neural_radiance_field.cuh:153-298).
//
// File = the whole network config JSON + a "snapshot" object, MessagePack-encoded (json::to_msgpack):
//   snapshot.version = 1, density_grid_size = 128, density_grid_binary = fp16 [(max_cascade+1) * 128^3] in Morton order,
//   nerf.aabb_scale, nerf.rgb.{rays_per_batch, measured_batch_size, measured_batch_size_before_compaction}, nerf.dataset
//   (json_binding.h:131-154), training_step, loss, aabb {min,max}, bounding_radius, and what tiny-cuda-nn's Trainer::serialize adds:
//   n_params, params_type ("__half" | "float"), params_binary (the inference weights), optional optimizer state.
// tiny-cuda-nn is absent from the reference tree, so the Trainer / optimizer keys follow its published layout from
// memory: byte compatibility with snapshots written by the CUDA build is UNPINNED until one can be read here.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "mini_json.h"
#include "ngp_hip.h"

namespace ngp {

struct NerfDataset;

uint16_t float_to_half_bits(float f);   // round to nearest even, like the device conversion
float half_bits_to_float(uint16_t h);

Json aabb_to_json(const NgpAabb& b);                               // json_binding.h:73-81
NgpAabb aabb_from_json(const Json& j);
Json mat_to_json(const float* col_major, int rows, int cols);      // json_binding.h:26-40: array of rows
void mat_from_json(const Json& j, float* col_major, int rows, int cols);
Json vec_to_json(const float* v, int n);
void vec_from_json(const Json& j, float* v, int n);

Json dataset_to_json(const NerfDataset& d);                        // json_binding.h:131-154
void dataset_from_json(const Json& j, NerfDataset& d);             // json_binding.h:156-201

// weights out of a "snapshot" object: fp16 as stored, or converted from a "float" snapshot
void snapshot_read_params(const Json& snapshot, std::vector<uint16_t>& params_fp16, std::vector<float>& params_fp32);
// density grid: fp16 Morton-ordered cells -> fp32
void snapshot_read_density_grid(const Json& snapshot, std::vector<float>& grid);

Json load_config_or_snapshot(const std::string& path);             // .json (comments + "parent" chain) or .msgpack (testbed.cu:120-145)

} // namespace ngp
