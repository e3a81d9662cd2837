// atomic_probe.hip — how fast are global atomics on MI355X, and does XCD locality matter?  (dev tool, not part of the product)
// hipcc --offload-arch=gfx950 -O3 tools/atomic_probe.hip -o /tmp/atomic_probe && /tmp/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

// MODE 0: random entries in [0, n_entries) (all blocks share the region)
// MODE 1: random entries inside a per-XCD slice of the region (slice = actual XCC id)
// MODE 2: like 0 but plain (non-atomic) stores, as a ceiling for scattered 4-byte writes
// MODE 3: like 0 with fp32 atomics
template <int MODE>
__global__ void __launch_bounds__(256) probe(h2* table, uint32_t n_entries, uint32_t per_thread, uint32_t seed) {
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t base = 0, range = n_entries;
	if (MODE == 1) { range = n_entries / 8; base = xcc_id() % 8 * range; }
	h2 v; v[0] = (_Float16)0.001f; v[1] = (_Float16)0.002f;
	for (uint32_t i = 0; i < per_thread; ++i) {
		const uint32_t idx = base + hash32(tid * 977u + i * 0x9e3779b9u + seed) % range;
		if (MODE == 2) table[idx] = v;
		else if (MODE == 3) atomicAdd((float*)(table + idx), 0.001f);
		else __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(table + idx), v);
	}
}

__global__ void __launch_bounds__(256) lds_probe(float* out, uint32_t per_thread, uint32_t seed) {
	__shared__ float t[16384];
	for (int i = threadIdx.x; i < 16384; i += 256) t[i] = 0;
	__syncthreads();
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	for (uint32_t i = 0; i < per_thread; ++i) atomicAdd(&t[hash32(tid * 977u + i * 0x9e3779b9u + seed) & 16383], 0.001f);
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = t[5];
}

int main() {
	const uint32_t blocks = 2048, per_thread = 64;
	const double n_ops = (double)blocks * 256 * per_thread;
	h2* table;
	CK(hipMalloc(&table, 64u << 20));
	CK(hipMemset(table, 0, 64u << 20));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const uint32_t sizes[] = {4096, 1u << 16, 1u << 19, 1u << 20, 6u << 20, 16u << 20};
	for (uint32_t n_entries : sizes) {
		for (int mode = 0; mode < 4; ++mode) {
			float best = 1e9f;
			for (int rep = 0; rep < 4; ++rep) {
				CK(hipEventRecord(e0));
				if (mode == 0) probe<0><<<blocks, 256>>>(table, n_entries, per_thread, rep);
				if (mode == 1) probe<1><<<blocks, 256>>>(table, n_entries, per_thread, rep);
				if (mode == 2) probe<2><<<blocks, 256>>>(table, n_entries, per_thread, rep);
				if (mode == 3) probe<3><<<blocks, 256>>>(table, n_entries, per_thread, rep);
				CK(hipEventRecord(e1));
				CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				if (rep > 0 && ms < best) best = ms;
			}
			const char* names[] = {"pk_f16 atomics, shared region", "pk_f16 atomics, per-XCD slice", "plain 4B stores", "f32 atomics, shared region"};
			printf("entries %9u (%6.1f MB)  %-32s %8.1f us  %7.2f Gops/s\n", n_entries, n_entries * 4.0 / 1e6, names[mode], best * 1e3, n_ops / best / 1e6);
		}
	}
	float* out; CK(hipMalloc(&out, blocks * 4));
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep) {
		CK(hipEventRecord(e0));
		lds_probe<<<blocks, 256>>>(out, per_thread, rep);
		CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		if (rep > 0 && ms < best) best = ms;
	}
	printf("LDS f32 atomics (64 KiB table per block)                    %8.1f us  %7.2f Gops/s\n", best * 1e3, n_ops / best / 1e6);
	return 0;
}
