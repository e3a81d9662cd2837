// probe.hip — a measuring stick, not a stage of the path: random 4-byte gathers from a table of the caller's choice, so that a benchmark can state the
// REQUEST-RATE roof of the box it runs on next to the byte roof (VERDICT r05 "next" #5: the network pass sits at the vector-L1 look-up rate, not at an HBM figure).
// bench.py times three launches of it with HIP events (table in L1 reach, one L2-sized slice per XCD, the hash table's 24 MiB shared by all XCDs: ~20 ms together)
// and reports the forward pass's gathers per second against them.  Same access pattern as tools/gather_probe.hip, which produced the figures quoted in DESIGN.md.
#include "ngp_device.cuh"

namespace ngp {

__device__ __forceinline__ uint32_t probe_hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t probe_xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }

// PER_XCD: every XCD gathers from its own 1 / 8 of the table (its L2 then holds the slice; nothing crosses the fabric once warm)
template <bool PER_XCD>
__global__ void __launch_bounds__(256) probe_gather_kernel(const uint32_t* __restrict__ table, uint32_t n_entries, uint32_t per_thread, uint32_t seed, uint32_t* __restrict__ sink) {
	const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
	uint32_t base = 0, range = n_entries;
	if (PER_XCD) { range = n_entries / 8u; base = probe_xcc_id() * range; }
	uint32_t acc = 0;
	for (uint32_t i = 0; i < per_thread; i += 8u) {
		uint32_t v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) v[u] = table[base + probe_hash32(tid * 977u + (i + u) * 0x9e3779b9u + seed) % range];
#pragma unroll
		for (int u = 0; u < 8; ++u) acc += v[u];
	}
	if (acc == 0x12345678u) sink[0] = acc;   // (never true for the tables the callers pass: keeps the loads alive)
}

}  // namespace ngp

extern "C" {

// n_blocks x 256 threads, per_thread (a multiple of 8) independent random 4-byte loads each from table[0 .. n_entries) (n_entries a multiple of 8); per_xcd != 0: each XCD
// confined to its own eighth.  Stream-ordered; the caller times it.  Returns the number of gathers the launch performs through *n_gathers_out (host).
int ngp_hip_probe_gather_rate(void* stream, const uint32_t* table, uint32_t n_entries, int per_xcd, uint32_t n_blocks, uint32_t per_thread, uint32_t seed, uint32_t* sink, uint64_t* n_gathers_out) {
	if (!table || !sink || n_entries < 8u || (n_entries & 7u) || !n_blocks || !per_thread || (per_thread & 7u)) {
		ngp::set_last_error("ngp_hip_probe_gather_rate: table / sink NULL, or n_entries / per_thread not positive multiples of 8", hipErrorInvalidValue); return -1;
	}
	if (per_xcd) hipLaunchKernelGGL(ngp::probe_gather_kernel<true>, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, table, n_entries, per_thread, seed, sink);
	else hipLaunchKernelGGL(ngp::probe_gather_kernel<false>, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, table, n_entries, per_thread, seed, sink);
	NGP_LAUNCH_CHECK("probe_gather_kernel");
	if (n_gathers_out) *n_gathers_out = (uint64_t)n_blocks * 256u * per_thread;
	return 0;
}

}  // extern "C"
