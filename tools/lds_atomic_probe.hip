// lds_atomic_probe.hip — LDS atomic throughput on MI355X by operation type and address pattern (dev tool)
// hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_probe.hip -o tools/lds_atomic_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// OP 0: ds_pk_add_f16, 1: ds_add_f32, 2: ds_add_u32, 3: plain ds_write_b32 (no atomic), 4: ds_add_u64
// PATTERN 0: random word in 128 KiB; 1: conflict-free (lane-linear, rotating base); 2: 8 lanes share an address; 3: the 64 lanes of a wave share one
template <int OP, int PATTERN>
__global__ void __launch_bounds__(1024) probe(uint32_t* out, uint32_t per_thread, uint32_t seed) {
	__shared__ uint32_t t[32768];
	for (int i = threadIdx.x; i < 32768; i += 1024) t[i] = 0;
	__syncthreads();
	const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
	h2 hv; hv[0] = (_Float16)0.001f; hv[1] = (_Float16)0.002f;
	for (uint32_t i = 0; i < per_thread; ++i) {
		uint32_t idx = PATTERN == 0 ? (hash32(tid * 977u + i * 0x9e3779b9u + seed) & 32767u) : PATTERN == 1 ? ((threadIdx.x + i * 1031u) & 32767u)
		             : PATTERN == 2 ? (hash32((tid >> 3) * 977u + i * 0x9e3779b9u + seed) & 32767u) : (hash32((tid >> 6) * 977u + i * 0x9e3779b9u + seed) & 32767u);
		if (OP == 0) __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2*)&t[idx], hv);
		else if (OP == 1) atomicAdd((float*)&t[idx], 0.001f);
		else if (OP == 2) atomicAdd(&t[idx], 3u);
		else if (OP == 3) t[idx] = i;
		else atomicAdd((unsigned long long*)&t[idx & ~1u], 3ull);
	}
	__syncthreads();
	if (threadIdx.x == 0) out[blockIdx.x] = t[5];
}

template <int OP, int PATTERN>
static float run(uint32_t* out, int blocks, uint32_t per_thread) {
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep) {
		hipEventRecord(e0);
		probe<OP, PATTERN><<<blocks, 1024>>>(out, per_thread, rep);
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	return best;
}

int main() {
	const int blocks = 512; const uint32_t per_thread = 256;
	const double n_ops = (double)blocks * 1024 * per_thread;
	uint32_t* out; CK(hipMalloc(&out, blocks * 4));
	const char* names[5] = {"ds_pk_add_f16", "ds_add_f32", "ds_add_u32", "ds_write_b32", "ds_add_u64"};
	printf("%-16s %14s %14s %14s %14s   (G lane-ops/s chip-wide, 512 blocks x 1024 threads)\n", "op", "random", "conflict-free", "8 share", "64 share");
	float r[5][4];
	r[0][0] = run<0, 0>(out, blocks, per_thread); r[0][1] = run<0, 1>(out, blocks, per_thread); r[0][2] = run<0, 2>(out, blocks, per_thread); r[0][3] = run<0, 3>(out, blocks, per_thread);
	r[1][0] = run<1, 0>(out, blocks, per_thread); r[1][1] = run<1, 1>(out, blocks, per_thread); r[1][2] = run<1, 2>(out, blocks, per_thread); r[1][3] = run<1, 3>(out, blocks, per_thread);
	r[2][0] = run<2, 0>(out, blocks, per_thread); r[2][1] = run<2, 1>(out, blocks, per_thread); r[2][2] = run<2, 2>(out, blocks, per_thread); r[2][3] = run<2, 3>(out, blocks, per_thread);
	r[3][0] = run<3, 0>(out, blocks, per_thread); r[3][1] = run<3, 1>(out, blocks, per_thread); r[3][2] = run<3, 2>(out, blocks, per_thread); r[3][3] = run<3, 3>(out, blocks, per_thread);
	r[4][0] = run<4, 0>(out, blocks, per_thread); r[4][1] = run<4, 1>(out, blocks, per_thread); r[4][2] = run<4, 2>(out, blocks, per_thread); r[4][3] = run<4, 3>(out, blocks, per_thread);
	for (int k = 0; k < 5; ++k) printf("%-16s %14.1f %14.1f %14.1f %14.1f\n", names[k], n_ops / r[k][0] * 1e-6, n_ops / r[k][1] * 1e-6, n_ops / r[k][2] * 1e-6, n_ops / r[k][3] * 1e-6);
	return 0;
}
