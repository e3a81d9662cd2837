// comm.hip — the collectives of the data-parallel training step behind the C ABI: one RCCL communicator per rank (one process per
// GPU, xGMI inside the node), stream-ordered all-reduces on the caller's HIP stream.  The reference has no multi-GPU path at all ("Can this
// codebase use multiple GPUs…? No.", README.md:239-241); these entry points are what its Trainer would call between backward and optimizer_step
// (src/testbed_nerf.cu:3331 / 2950) and around NerfCounters::update_after_training (2870-2894).
//
// RCCL is bound at run time (dlopen): libngp_hip.so itself needs libamdhip64 only, and a process that already carries an RCCL — PyTorch bundles
// its own next to its own HIP runtime — must keep using THAT copy (one HIP runtime per process), so a loaded librccl is looked up
// first and /opt/rocm/lib is the fallback.
#include <dlfcn.h>
#include <link.h>
#include <string.h>
#include <string>

#include "ngp_device.cuh"

namespace {

typedef struct { char internal[128]; } RcclUniqueId;   // ncclUniqueId (rccl.h:43)
typedef void* RcclComm;
enum { RCCL_SUM = 0, RCCL_FLOAT16 = 6, RCCL_FLOAT32 = 7, RCCL_FLOAT64 = 8 };   // ncclRedOp_t / ncclDataType_t values (rccl.h)

struct RcclApi {
	void* handle = nullptr;
	int (*GetUniqueId)(RcclUniqueId*) = nullptr;
	int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
	int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
	int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
	int (*ReduceScatter)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
	int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*CommCount)(RcclComm, int*) = nullptr;
	int (*CommDestroy)(RcclComm) = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	std::string where;
};

struct Loaded { std::string rccl, hip_dir; };
int find_loaded(struct dl_phdr_info* info, size_t, void* out) {
	Loaded* l = (Loaded*)out;
	if (!info->dlpi_name) return 0;
	if (strstr(info->dlpi_name, "librccl")) l->rccl = info->dlpi_name;
	if (const char* p = strstr(info->dlpi_name, "libamdhip64")) l->hip_dir = std::string(info->dlpi_name, (size_t)(p - info->dlpi_name));
	return 0;
}

RcclApi* rccl() {
	static RcclApi api;
	static bool tried = false;
	if (tried) return api.handle ? &api : nullptr;
	tried = true;
	// 0. NGP_RCCL_LIBRARY: the operator's choice (another RCCL build; tests/loopback's several-ranks-on-one-GPU stand-in) — used or the binding fails, never skipped;
	// 1. an RCCL this process already carries; 2. the one that sits NEXT TO the HIP runtime in use (PyTorch ships librccl.so beside its own
	// libamdhip64 and loads it lazily: binding /opt/rocm's copy instead would pull a second HIP runtime into the process); 3. the system's
	if (const char* forced = getenv("NGP_RCCL_LIBRARY")) {
		if (*forced) {
			api.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
			if (!api.handle) return nullptr;
			api.where = forced;
		}
	}
	Loaded l;
	if (!api.handle) dl_iterate_phdr(find_loaded, &l);
	const std::string beside = l.hip_dir.empty() ? std::string() : l.hip_dir + "librccl.so", beside1 = l.hip_dir.empty() ? std::string() : l.hip_dir + "librccl.so.1";
	const char* candidates[] = {l.rccl.empty() ? nullptr : l.rccl.c_str(), beside.empty() ? nullptr : beside.c_str(), beside1.empty() ? nullptr : beside1.c_str(),
	                            "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	for (const char* c : candidates) {
		if (api.handle) break;
		if (!c) continue;
		api.handle = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
		if (api.handle) { api.where = c; break; }
	}
	if (!api.handle) return nullptr;
	api.GetUniqueId = (int (*)(RcclUniqueId*))dlsym(api.handle, "ncclGetUniqueId");
	api.CommInitRank = (int (*)(RcclComm*, int, RcclUniqueId, int))dlsym(api.handle, "ncclCommInitRank");
	api.AllReduce = (int (*)(const void*, void*, size_t, int, int, RcclComm, hipStream_t))dlsym(api.handle, "ncclAllReduce");
	api.AllGather = (int (*)(const void*, void*, size_t, int, RcclComm, hipStream_t))dlsym(api.handle, "ncclAllGather");
	api.ReduceScatter = (int (*)(const void*, void*, size_t, int, int, RcclComm, hipStream_t))dlsym(api.handle, "ncclReduceScatter");
	api.Send = (int (*)(const void*, size_t, int, int, RcclComm, hipStream_t))dlsym(api.handle, "ncclSend");
	api.Recv = (int (*)(void*, size_t, int, int, RcclComm, hipStream_t))dlsym(api.handle, "ncclRecv");
	api.GroupStart = (int (*)())dlsym(api.handle, "ncclGroupStart");
	api.GroupEnd = (int (*)())dlsym(api.handle, "ncclGroupEnd");
	api.CommCount = (int (*)(RcclComm, int*))dlsym(api.handle, "ncclCommCount");
	api.CommDestroy = (int (*)(RcclComm))dlsym(api.handle, "ncclCommDestroy");
	api.GetErrorString = (const char* (*)(int))dlsym(api.handle, "ncclGetErrorString");
	if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.AllGather || !api.ReduceScatter || !api.CommDestroy) { dlclose(api.handle); api.handle = nullptr; return nullptr; }
	return &api;
}

int fail(const char* what, int rc) {
	RcclApi* a = rccl();
	std::string msg = std::string(what) + ": " + (a && a->GetErrorString ? a->GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
	ngp::set_last_error(msg.c_str(), hipErrorUnknown);
	return -1;
}

struct Comm { RcclComm comm = nullptr; int rank = 0, world = 1; };

}  // namespace

namespace ngp {
// out[i] = half( sum over q = 0 .. world - 1, in that order, of float(slices[q][i]) ): fp32 accumulation, ONE fp16 rounding — 8 elements (16 bytes) per thread and slice
__global__ void __launch_bounds__(256) sum_slices_f16_kernel(uint32_t world, uint32_t count, const half_t* __restrict__ slices, half_t* __restrict__ out) {
	typedef _Float16 h8v __attribute__((ext_vector_type(8)));
	const uint32_t i = (blockIdx.x * 256u + threadIdx.x) * 8u;
	if (i >= count) return;
	if (i + 8u <= count) {
		float acc[8];
		const h8v v0 = *(const h8v*)(slices + i);
#pragma unroll
		for (int k = 0; k < 8; ++k) acc[k] = (float)v0[k];
		for (uint32_t q = 1; q < world; ++q) {
			const h8v v = *(const h8v*)(slices + (size_t)q * count + i);
#pragma unroll
			for (int k = 0; k < 8; ++k) acc[k] += (float)v[k];
		}
		h8v r;
#pragma unroll
		for (int k = 0; k < 8; ++k) r[k] = (half_t)acc[k];
		*(h8v*)(out + i) = r;
	} else {
		for (uint32_t e = i; e < count; ++e) {
			float acc = (float)slices[e];
			for (uint32_t q = 1; q < world; ++q) acc += (float)slices[(size_t)q * count + e];
			out[e] = (half_t)acc;
		}
	}
}
}  // namespace ngp

extern "C" {

// `slices`: world x count fp16 (slice q = rank q's contribution, 16-byte aligned, count a multiple of 8 or any with an unaligned tail); out: count fp16
int ngp_hip_sum_slices_f16(void* stream, uint32_t world, uint32_t count, const uint16_t* slices, uint16_t* out) {
	if (count == 0) return 0;
	if (world == 0 || !slices || !out || ((uintptr_t)slices & 15u) || ((uintptr_t)out & 15u) || (world > 1 && (count & 7u))) {
		ngp::set_last_error("ngp_hip_sum_slices_f16: world >= 1, 16-byte aligned buffers, count a multiple of 8 when world > 1", hipErrorInvalidValue); return -1;
	}
	hipLaunchKernelGGL(ngp::sum_slices_f16_kernel, dim3((count + 2047u) / 2048u), dim3(256), 0, (hipStream_t)stream, world, count, (const ngp::half_t*)slices, (ngp::half_t*)out);
	NGP_LAUNCH_CHECK("sum_slices_f16_kernel");
	return 0;
}

int ngp_rccl_available(void) { return rccl() ? 1 : 0; }

int ngp_rccl_get_unique_id(uint8_t* out128) {
	RcclApi* a = rccl();
	if (!a) { ngp::set_last_error("ngp_rccl_get_unique_id: librccl could not be loaded", hipErrorNotSupported); return -1; }
	RcclUniqueId id;
	const int rc = a->GetUniqueId(&id);
	if (rc) return fail("ncclGetUniqueId", rc);
	memcpy(out128, id.internal, 128);
	return 0;
}

void* ngp_rccl_init(int rank, int world_size, const uint8_t* unique_id128) {
	RcclApi* a = rccl();
	if (!a) { ngp::set_last_error("ngp_rccl_init: librccl could not be loaded", hipErrorNotSupported); return nullptr; }
	if (world_size < 1 || rank < 0 || rank >= world_size || !unique_id128) { ngp::set_last_error("ngp_rccl_init: bad rank / world_size / id", hipErrorInvalidValue); return nullptr; }
	RcclUniqueId id;
	memcpy(id.internal, unique_id128, 128);
	Comm* c = new Comm();
	c->rank = rank; c->world = world_size;
	const int rc = a->CommInitRank(&c->comm, world_size, id, rank);   // collective: every rank of the job calls it with the same id, its own device current
	if (rc) { fail("ncclCommInitRank", rc); delete c; return nullptr; }
	return c;
}

static int all_reduce(void* comm, void* stream, void* buf, uint64_t count, int dtype, const char* who) {
	RcclApi* a = rccl();
	Comm* c = (Comm*)comm;
	if (!a || !c) { ngp::set_last_error(who, hipErrorInvalidValue); return -1; }
	if (count == 0) return 0;
	const int rc = a->AllReduce(buf, buf, (size_t)count, dtype, RCCL_SUM, c->comm, (hipStream_t)stream);   // in place, ordered on the caller's stream
	return rc ? fail(who, rc) : 0;
}

int ngp_rccl_allreduce_grads(void* comm, void* stream, uint16_t* grads_f16, uint64_t n_params) { return all_reduce(comm, stream, grads_f16, n_params, RCCL_FLOAT16, "ngp_rccl_allreduce_grads"); }
int ngp_rccl_allreduce_f32(void* comm, void* stream, float* values, uint64_t count) { return all_reduce(comm, stream, values, count, RCCL_FLOAT32, "ngp_rccl_allreduce_f32"); }
int ngp_rccl_allreduce_counters(void* comm, void* stream, double* values, uint64_t count) { return all_reduce(comm, stream, values, count, RCCL_FLOAT64, "ngp_rccl_allreduce_counters"); }

// in place: rank r's chunk sits at buf + r * count_per_rank elements before the call, every rank holds all chunks after it
static int all_gather(void* comm, void* stream, void* buf, uint64_t count_per_rank, int dtype, size_t elem_bytes, const char* who) {
	RcclApi* a = rccl();
	Comm* c = (Comm*)comm;
	if (!a || !c) { ngp::set_last_error(who, hipErrorInvalidValue); return -1; }
	if (count_per_rank == 0) return 0;
	const int rc = a->AllGather((const char*)buf + (size_t)c->rank * count_per_rank * elem_bytes, buf, (size_t)count_per_rank, dtype, c->comm, (hipStream_t)stream);
	return rc ? fail(who, rc) : 0;
}
int ngp_rccl_allgather_f32(void* comm, void* stream, float* buf, uint64_t count_per_rank) { return all_gather(comm, stream, buf, count_per_rank, RCCL_FLOAT32, 4, "ngp_rccl_allgather_f32"); }
int ngp_rccl_allgather_f16(void* comm, void* stream, uint16_t* buf, uint64_t count_per_rank) { return all_gather(comm, stream, buf, count_per_rank, RCCL_FLOAT16, 2, "ngp_rccl_allgather_f16"); }

// sum over the ranks of `in` (world x count_per_rank elements); rank r receives elements [r * count_per_rank, (r + 1) * count_per_rank) of the sum in out
int ngp_rccl_reduce_scatter_f32(void* comm, void* stream, const float* in, float* out, uint64_t count_per_rank) {
	RcclApi* a = rccl();
	Comm* c = (Comm*)comm;
	if (!a || !c) { ngp::set_last_error("ngp_rccl_reduce_scatter_f32", hipErrorInvalidValue); return -1; }
	if (count_per_rank == 0) return 0;
	const int rc = a->ReduceScatter(in, out, (size_t)count_per_rank, RCCL_FLOAT32, RCCL_SUM, c->comm, (hipStream_t)stream);
	return rc ? fail("ngp_rccl_reduce_scatter_f32", rc) : 0;
}

// The gradient exchange with fp16 ON THE WIRE (round 5): rank r sends slice q of `send` (world x count_per_rank fp16) to rank q and receives rank q's slice r into slice q
// of `recv` — point-to-point over the direct xGMI links of a fully connected node, (world - 1) / world x 2 bytes per parameter per rank instead of the fp32
// reduce-scatter's 4.  No arithmetic on the wire: the owner of a slice then sums the world fp16 vectors in RANK ORDER in fp32 and rounds once
// (ngp_hip_sum_slices_f16) — a summation order that does not depend on the collective library's ring / tree choice.  The own slice is copied device to device.
int ngp_rccl_alltoall_f16(void* comm, void* stream, const uint16_t* send, uint16_t* recv, uint64_t count_per_rank) {
	RcclApi* a = rccl();
	Comm* c = (Comm*)comm;
	if (!a || !c) { ngp::set_last_error("ngp_rccl_alltoall_f16", hipErrorInvalidValue); return -1; }
	if (count_per_rank == 0) return 0;
	if (hipMemcpyAsync(recv + (size_t)c->rank * count_per_rank, send + (size_t)c->rank * count_per_rank, (size_t)count_per_rank * 2u, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
		ngp::set_last_error("ngp_rccl_alltoall_f16: device copy of the own slice failed", hipErrorUnknown); return -1;
	}
	if (c->world == 1) return 0;
	if (!a->Send || !a->Recv || !a->GroupStart || !a->GroupEnd) { ngp::set_last_error("ngp_rccl_alltoall_f16: this librccl has no ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd", hipErrorNotSupported); return -1; }
	int rc = a->GroupStart();
	if (rc) return fail("ncclGroupStart", rc);
	for (int q = 0; q < c->world && !rc; ++q) {
		if (q == c->rank) continue;
		rc = a->Send(send + (size_t)q * count_per_rank, (size_t)count_per_rank, RCCL_FLOAT16, q, c->comm, (hipStream_t)stream);
		if (!rc) rc = a->Recv(recv + (size_t)q * count_per_rank, (size_t)count_per_rank, RCCL_FLOAT16, q, c->comm, (hipStream_t)stream);
	}
	const int rc_end = a->GroupEnd();
	if (rc) return fail("ncclSend / ncclRecv (ngp_rccl_alltoall_f16)", rc);
	return rc_end ? fail("ncclGroupEnd (ngp_rccl_alltoall_f16)", rc_end) : 0;
}

int ngp_rccl_comm_size(void* comm) {   // ncclCommCount: what the communicator itself says (bench.py prints it next to WORLD_SIZE)
	RcclApi* a = rccl();
	Comm* c = (Comm*)comm;
	if (!a || !c) return -1;
	int n = c->world;
	if (a->CommCount && a->CommCount(c->comm, &n) != 0) return -1;
	return n;
}
int ngp_rccl_comm_rank(void* comm) { Comm* c = (Comm*)comm; return c ? c->rank : -1; }

int ngp_rccl_finalize(void* comm) {
	Comm* c = (Comm*)comm;
	if (!c) return 0;
	RcclApi* a = rccl();
	int rc = 0;
	if (a && c->comm) rc = a->CommDestroy(c->comm);
	delete c;
	return rc ? fail("ncclCommDestroy", rc) : 0;
}

}  // extern "C"
