// Development knobs: sweeps and timing ablations of tools/*.  The PRODUCT library (lib/libngp_hip.so) is built without NGP_DEV_KNOBS: every knob folds to its
// default at compile time and no launch path reads the environment.  `python blender-ngp_amd/build.py --dev` builds the same sources with -DNGP_DEV_KNOBS into
// lib_dev/libngp_hip.so (git-ignored, ships with gpurun); a tool opts in with NGP_HIP_LIBRARY_DIR=blender-ngp_amd/lib_dev (capi.py) + LD_LIBRARY_PATH for pyngp.
#pragma once
#include <cstdint>
#include <cstdlib>

#ifdef NGP_DEV_KNOBS
static inline uint32_t ngp_dev_knob_u32(const char* name, uint32_t dflt) { const char* v = getenv(name); return v ? (uint32_t)strtoul(v, nullptr, 0) : dflt; }
static inline bool ngp_dev_knob_set(const char* name) { return getenv(name) != nullptr; }
#else
static inline constexpr uint32_t ngp_dev_knob_u32(const char*, uint32_t dflt) { return dflt; }
static inline constexpr bool ngp_dev_knob_set(const char*) { return false; }
#endif
