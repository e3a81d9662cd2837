// fetch_calibration.hip — what do rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ* report for the access patterns of this code base, and how many bytes does such an access
// really move?  (VERDICT r05 "weak" #3 / "next" #5a: tools/pmc_traffic.py doubled FETCH_SIZE for every kernel, a factor MI355X_MICROARCH.md gives for wide coalesced
// streams only.)  Five kernels over a 2 GiB table — far beyond the 256 MiB Infinity Cache, so every access is a memory access:
//   stream16     16 B per lane, coalesced: the guide's calibration pattern, 2 GiB of known traffic
//   line128_4B   one 4-byte load per 128-byte line, every line once (lanes of a wave on consecutive lines)
//   line64_4B    one 4-byte load per 64-byte half line, every half line once
//   random_4B    4-byte loads at hashed addresses (the fused forward pass's corner gathers), 2^26 of them
//   random_8B    8-byte aligned pair loads at hashed addresses (the encoder's x-pair gathers), 2^26 of them
// Run plainly it prints each kernel's best time: line128_4B against stream16 says whether a 4-byte touch costs the memory system 128 bytes or 64 (it runs at the
// stream's pace per LINE or twice as fast).  Under `rocprofv3 --pmc ...` (tools/gpu_r06_calib.sh) the counters per dispatch give requests and "bytes" per access.
// dev tool, not part of the product:  hipcc --offload-arch=gfx950 -O3 tools/fetch_calibration.hip -o /tmp/fetch_calibration
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void __launch_bounds__(256) stream16(const uint4* __restrict__ t, uint64_t n16, uint32_t* out) {
	uint32_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) { const uint4 v = t[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
	if (acc == 0x12345678u) out[0] = acc;
}
template <int STRIDE_BYTES>
__global__ void __launch_bounds__(256) line_touch_4B(const uint32_t* __restrict__ t, uint64_t n_lines, uint32_t* out) {
	uint32_t acc = 0;
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_lines; i += (uint64_t)gridDim.x * 256) acc += t[i * (STRIDE_BYTES / 4)];
	if (acc == 0x12345678u) out[0] = acc;
}
template <int WIDTH>
__global__ void __launch_bounds__(256) random_gather(const uint32_t* __restrict__ t, uint32_t n_entries_mask, uint32_t per_thread, uint32_t* out) {
	const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
	uint32_t acc = 0;
	for (uint32_t i = 0; i < per_thread; i += 8) {
		uint32_t v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			uint32_t idx = hash32(tid * 977u + (i + u) * 0x9e3779b9u) & n_entries_mask;
			if (WIDTH == 2) { idx &= ~1u; const uint2 p = *(const uint2*)(t + idx); v[u] = p.x ^ p.y; } else v[u] = t[idx];
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) acc += v[u];
	}
	if (acc == 0x12345678u) out[0] = acc;
}

int main() {
	const uint64_t bytes = 2ull << 30;
	uint32_t *table, *out;
	CK(hipMalloc(&table, bytes)); CK(hipMemset(table, 1, bytes)); CK(hipMalloc(&out, 64));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const uint32_t n_gather_blocks = 4096, per_thread = 64;   // 2^26 gathers
	const char* names[5] = {"stream16", "line128_4B", "line64_4B", "random_4B", "random_8B"};
	const double units[5] = {(double)bytes / 128.0, (double)bytes / 128.0, (double)bytes / 64.0, (double)n_gather_blocks * 256 * per_thread, (double)n_gather_blocks * 256 * per_thread};
	printf("{\"table_bytes\": %llu", (unsigned long long)bytes);
	for (int k = 0; k < 5; ++k) {
		float best = 1e30f;
		for (int rep = 0; rep < 3; ++rep) {
			CK(hipEventRecord(e0));
			if (k == 0) stream16<<<8192, 256>>>((const uint4*)table, bytes / 16, out);
			if (k == 1) line_touch_4B<128><<<8192, 256>>>(table, bytes / 128, out);
			if (k == 2) line_touch_4B<64><<<8192, 256>>>(table, bytes / 64, out);
			if (k == 3) random_gather<1><<<n_gather_blocks, 256>>>(table, (uint32_t)(bytes / 4 - 1), per_thread, out);
			if (k == 4) random_gather<2><<<n_gather_blocks, 256>>>(table, (uint32_t)(bytes / 4 - 1), per_thread, out);
			CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		printf(", \"%s\": {\"accesses\": %.0f, \"best_ms\": %.4f, \"G_accesses_per_s\": %.2f}", names[k], units[k], best, units[k] / best * 1e-6);
	}
	printf("}\n");
	return 0;
}
