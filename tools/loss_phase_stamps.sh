#!/bin/bash
# dev (no GPU needed): builds blender-ngp_amd/lib_t/libngp_hip.so = the kernel library with cycle stamps in compute_loss_kernel (start, after the per-ray set-up, after pass 1,
# after the compaction barrier, end) and ngp_hip_debug_loss_timing() to read them; tools/loss_phase_probe.py (GPU box, LD_LIBRARY_PATH=blender-ngp_amd/lib_t) prints the phases.
cd "$(dirname "$0")/../blender-ngp_amd" || exit 1
python3 - <<'PY'
s = open("csrc/loss.hip").read()
def rep(old, new):
    global s
    assert s.count(old) == 1, old[:60]
    s = s.replace(old, new)
rep("__global__ void __launch_bounds__(LOSS_RAYS_PER_BLOCK * 64) compute_loss_kernel(const LossArgs a) {",
    "__device__ unsigned long long g_loss_t[8192 * 8];\n__global__ void __launch_bounds__(LOSS_RAYS_PER_BLOCK * 64) compute_loss_kernel(const LossArgs a) {\n"
    "\tconst unsigned long long t_start = __builtin_readcyclecounter();\n\tunsigned long long t_setup = 0, t_p1 = 0, t_bar = 0, t_p2 = 0;")
rep("\t// ---- pass 1: transmittance, ray colour", "\tt_setup = __builtin_readcyclecounter();\n\t// ---- pass 1: transmittance, ray colour")
rep("\t// ---- compaction slots: one atomic per workgroup (1434)", "\tt_p1 = __builtin_readcyclecounter();\n\t// ---- compaction slots: one atomic per workgroup (1434)")
rep("\t// (slots that produce no loss get a zero:", "\tt_bar = __builtin_readcyclecounter();\n\t// (slots that produce no loss get a zero:")
rep("\t\tdepth_carry = wave_last(depth_ray2);\n\t}\n}",
    "\t\tdepth_carry = wave_last(depth_ray2);\n\t}\n\tt_p2 = __builtin_readcyclecounter();\n"
    "\tif (lane == 0 && i < 8192u) { unsigned long long* o = g_loss_t + (size_t)i * 8; o[0] = t_start; o[1] = t_setup; o[2] = t_p1; o[3] = t_bar; o[4] = t_p2; o[5] = numsteps; o[6] = compacted; o[7] = a.rng.state; }\n}\n"
    "extern \"C\" __attribute__((visibility(\"default\"))) int ngp_hip_debug_loss_timing(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_loss_t), sizeof(unsigned long long) * 8192 * 8, 0, hipMemcpyDeviceToHost); }")
open("/tmp/loss_stamped.hip", "w").write(s)
PY
mkdir -p lib_t
cp /tmp/loss_stamped.hip csrc/_loss_stamped.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -I../include -c csrc/_loss_stamped.hip -o build/loss_t.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/density_grid.o build/train_samples.o build/network.o build/loss_t.o build/render.o build/multi_render.o build/comm.o -o lib_t/libngp_hip.so -ldl && echo "built lib_t"
rm -f csrc/_loss_stamped.hip build/loss_t.o
