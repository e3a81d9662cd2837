// gather_probe.hip — ceilings for random small gathers on MI355X: 4-byte / 8-byte loads from tables resident in L1 / L2 / MALL / HBM,
// all blocks sharing the table (MODE 0) or each XCD confined to its own 1/8 slice (MODE 1).  (dev tool, not part of the product)
// hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

// WIDTH 1: 4-byte loads; 2: 8-byte loads (aligned pairs).  UNROLL independent loads in flight per lane.
template <int MODE, int WIDTH>
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ table, uint32_t n_entries, uint32_t per_thread, uint32_t seed, uint32_t* out) {
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t base = 0, range = n_entries;
	if (MODE == 1) { range = n_entries / 8; base = xcc_id() % 8 * range; }
	uint32_t acc = 0;
	for (uint32_t i = 0; i < per_thread; i += 8) {
		uint32_t v[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			uint32_t idx = base + hash32(tid * 977u + (i + u) * 0x9e3779b9u + seed) % range;
			if (WIDTH == 2) { idx &= ~1u; const uint2 t = *(const uint2*)(table + idx); v[u] = t.x ^ t.y; }
			else v[u] = table[idx];
		}
#pragma unroll
		for (int u = 0; u < 8; ++u) acc += v[u];
	}
	if (acc == 0x12345678u) out[tid] = acc;
}

int main() {
	const uint32_t blocks = 8192, per_thread = 64;
	const double n_ops = (double)blocks * 256 * per_thread;
	uint32_t *table, *out;
	CK(hipMalloc(&table, 1u << 30));
	CK(hipMemset(table, 1, 1u << 30));
	CK(hipMalloc(&out, blocks * 256 * 4));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const uint32_t sizes[] = {2048, 1u << 16, 1u << 19, 1u << 20, 1u << 22, 6u << 20, 1u << 25, 1u << 28};  // entries of 4 B
	printf("%12s %10s %10s %10s %10s   (G gathers/s, chip-wide)\n", "table bytes", "4B shared", "4B perXCD", "8B shared", "8B perXCD");
	for (uint32_t n_entries : sizes) {
		printf("%12u", n_entries * 4);
		for (int v = 0; v < 4; ++v) {
			float best = 1e9f;
			for (int rep = 0; rep < 4; ++rep) {
				CK(hipEventRecord(e0));
				if (v == 0) probe<0, 1><<<blocks, 256>>>(table, n_entries, per_thread, rep, out);
				if (v == 1) probe<1, 1><<<blocks, 256>>>(table, n_entries, per_thread, rep, out);
				if (v == 2) probe<0, 2><<<blocks, 256>>>(table, n_entries, per_thread, rep, out);
				if (v == 3) probe<1, 2><<<blocks, 256>>>(table, n_entries, per_thread, rep, out);
				CK(hipEventRecord(e1));
				CK(hipEventSynchronize(e1));
				float ms; CK(hipEventElapsedTime(&ms, e0, e1));
				if (ms < best) best = ms;
			}
			printf(" %10.1f", n_ops / best * 1e-6);
		}
		printf("\n");
	}
	return 0;
}
