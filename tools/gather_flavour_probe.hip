// gather_flavour_probe.hip — does a cache-policy bit change what a random 4- / 8-byte gather costs on MI355X?  A gather that misses the vector L1 pulls a whole
// L1 line over the 64 B/clk L2->L1 path; if a policy bit makes the L2 return only what was asked for, the L2-hit gather ceiling (~300 G/s) moves.
// Tables: 2 MiB (resident in every XCD's L2) and 24 MiB (lives in the Infinity Cache).  (dev tool, not part of the product)
// hipcc --offload-arch=gfx950 -O3 tools/gather_flavour_probe.hip -o /tmp/gfp && /tmp/gfp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

#define LOAD4(FLAV) asm volatile("global_load_dword %0, %1, off " FLAV : "=&v"(v[u]) : "v"(p) : "memory")
#define LOAD8(FLAV) asm volatile("global_load_dwordx2 %0, %1, off " FLAV : "=&v"(w[u]) : "v"(p) : "memory")

template <int FLAV, int WIDTH>
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ table, uint32_t n_entries, uint32_t per_thread, uint32_t seed, uint32_t* out) {
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t acc = 0;
	for (uint32_t i = 0; i < per_thread; i += 8) {
		uint32_t v[8]; uint2 w[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) {
			uint32_t idx = hash32(tid * 977u + (i + u) * 0x9e3779b9u + seed) % n_entries;
			if (WIDTH == 2) idx &= ~1u;
			const uint32_t* p = table + idx;
			if (WIDTH == 1) {
				if (FLAV == 0) LOAD4(""); else if (FLAV == 1) LOAD4("sc0"); else if (FLAV == 2) LOAD4("sc1"); else if (FLAV == 3) LOAD4("sc0 sc1");
				else if (FLAV == 4) LOAD4("nt"); else if (FLAV == 5) LOAD4("sc1 nt"); else LOAD4("sc0 sc1 nt");
			} else {
				if (FLAV == 0) LOAD8(""); else if (FLAV == 1) LOAD8("sc0"); else if (FLAV == 2) LOAD8("sc1"); else if (FLAV == 3) LOAD8("sc0 sc1");
				else if (FLAV == 4) LOAD8("nt"); else if (FLAV == 5) LOAD8("sc1 nt"); else LOAD8("sc0 sc1 nt");
			}
		}
		// the loads are invisible to the compiler's vmcnt bookkeeping: every result register passes through the wait, so that no use (and no reuse) can move above it
		if (WIDTH == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
		else asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) :: "memory");
#pragma unroll
		for (int u = 0; u < 8; ++u) acc += WIDTH == 1 ? v[u] : (w[u].x ^ w[u].y);
	}
	if (acc == 0x12345678u) out[tid] = acc;
}

template <int FLAV, int WIDTH>
static float run(const uint32_t* table, uint32_t n_entries, uint32_t blocks, uint32_t per_thread, uint32_t* out, hipEvent_t e0, hipEvent_t e1) {
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep) {
		hipEventRecord(e0);
		probe<FLAV, WIDTH><<<blocks, 256>>>(table, n_entries, per_thread, rep, out);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	return best;
}

int main() {
	const uint32_t blocks = 8192, per_thread = 64;
	const double n_ops = (double)blocks * 256 * per_thread;
	uint32_t *table, *out;
	CK(hipMalloc(&table, 1u << 28));
	CK(hipMemset(table, 1, 1u << 28));
	CK(hipMalloc(&out, blocks * 256 * 4));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const char* names[7] = {"plain", "sc0", "sc1", "sc0 sc1", "nt", "sc1 nt", "sc0 sc1 nt"};
	const uint32_t sizes[] = {4096, 1u << 19, 6u << 20, 1u << 26};  // entries of 4 B: 16 KiB (L1), 2 MiB (every L2), 24 MiB (Infinity Cache), 256 MiB
	printf("G gathers/s, chip-wide; columns: table bytes\n%-12s", "flavour");
	for (uint32_t n : sizes) printf(" %9uB/4 %9uB/8", n * 4, n * 4);
	printf("\n");
	for (int f = 0; f < 7; ++f) {
		printf("%-12s", names[f]);
		for (uint32_t n : sizes) {
			float a = 0, b = 0;
			switch (f) {
				case 0: a = run<0, 1>(table, n, blocks, per_thread, out, e0, e1); b = run<0, 2>(table, n, blocks, per_thread, out, e0, e1); break;
				case 1: a = run<1, 1>(table, n, blocks, per_thread, out, e0, e1); b = run<1, 2>(table, n, blocks, per_thread, out, e0, e1); break;
				case 2: a = run<2, 1>(table, n, blocks, per_thread, out, e0, e1); b = run<2, 2>(table, n, blocks, per_thread, out, e0, e1); break;
				case 3: a = run<3, 1>(table, n, blocks, per_thread, out, e0, e1); b = run<3, 2>(table, n, blocks, per_thread, out, e0, e1); break;
				case 4: a = run<4, 1>(table, n, blocks, per_thread, out, e0, e1); b = run<4, 2>(table, n, blocks, per_thread, out, e0, e1); break;
				case 5: a = run<5, 1>(table, n, blocks, per_thread, out, e0, e1); b = run<5, 2>(table, n, blocks, per_thread, out, e0, e1); break;
				default: a = run<6, 1>(table, n, blocks, per_thread, out, e0, e1); b = run<6, 2>(table, n, blocks, per_thread, out, e0, e1); break;
			}
			printf(" %11.1f %11.1f", n_ops / a * 1e-6, n_ops / b * 1e-6);
		}
		printf("\n");
	}
	return 0;
}
