// rccl_loopback.hip — TEST INFRASTRUCTURE, never shipped: the handful of nccl* entry points that blender-ngp_amd/csrc/comm.hip resolves with dlsym, implemented for
// SEVERAL PROCESSES ON ONE GPU.  Real RCCL refuses a communicator with two ranks on the same device ("duplicate GPU"), so on a one-GPU box the N > 1 branches of
// the data-parallel training step (host/testbed.cpp: optimizer_step_sharded, the stale-state flags, the row-sharded render) could never run on a device.  With
// NGP_RCCL_LIBRARY pointing here they do: product code is the same binary, only the library behind the nccl* names differs.
//
// How a collective works here (nothing is borrowed from RCCL's algorithms — this is a correctness vehicle, not a transport):
//   * ncclCommInitRank: every rank maps one POSIX shared-memory block named by the unique id, hipMalloc's a staging buffer, publishes its hipIpcMemHandle in the
//     block and opens the handles of its peers (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0);
//   * a collective = copy the send data into the own staging buffer on the caller's stream, drain the stream, barrier in shared memory, read the peers' staging
//     buffers (device copies, or a reduction kernel summing in RANK ORDER — fp32 accumulation and one rounding for fp16), drain, barrier.  Synchronous on the host:
//     a legal realisation of stream-ordered semantics for callers that issue their collectives in the same order on every rank;
//   * every rank posts {sequence number, kind, count, dtype} before the first barrier and compares with all peers behind it: ranks that disagree about a collective
//     get ncclInvalidUsage and a message on stderr instead of silently exchanging garbage;
//   * every wait has a time-out (NGP_LOOPBACK_TIMEOUT_S, default 60): a missing rank ends in ncclSystemError, not in a hung GPU box.
// Grouped ncclSend / ncclRecv (the fp16 all-to-all of the gradient exchange): executed at ncclGroupEnd, which EVERY rank of the communicator must reach.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

enum { OK = 0, UNHANDLED = 1, SYSTEM_ERROR = 2, INTERNAL_ERROR = 3, INVALID_ARGUMENT = 4, INVALID_USAGE = 5 };   // ncclResult_t
enum { DT_I8 = 0, DT_U8 = 1, DT_I32 = 2, DT_U32 = 3, DT_I64 = 4, DT_U64 = 5, DT_F16 = 6, DT_F32 = 7, DT_F64 = 8 };  // ncclDataType_t
enum { KIND_ALLREDUCE = 1, KIND_ALLGATHER = 2, KIND_REDUCESCATTER = 3, KIND_GROUP = 4 };
constexpr int MAX_WORLD = 16, MAX_P2P = 64;
constexpr uint64_t MAGIC = 0x6c6f6f706261636bull;   // "loopback"

typedef std::chrono::steady_clock Clock;
double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }
double timeout_s() { const char* e = getenv("NGP_LOOPBACK_TIMEOUT_S"); return e ? atof(e) : 60.0; }
size_t staging_bytes() { const char* e = getenv("NGP_LOOPBACK_STAGING_MB"); return (size_t)(e ? atoi(e) : 128) << 20; }

size_t dtype_size(int dt) {
	switch (dt) { case DT_I8: case DT_U8: return 1; case DT_F16: return 2; case DT_I32: case DT_U32: case DT_F32: return 4; case DT_I64: case DT_U64: case DT_F64: return 8; default: return 0; }
}

struct P2P { uint32_t peer; uint32_t pad; uint64_t offset, bytes; };
struct RankBlock {   // one per rank, written by that rank only
	hipIpcMemHandle_t handle;           // 64 bytes
	volatile uint64_t ready;            // staging published
	volatile uint64_t seq, kind, count, dtype;   // the collective this rank is in
	volatile uint64_t n_sends;
	P2P sends[MAX_P2P];
	uint8_t pad[64];
};
struct Shared {
	volatile uint64_t magic;
	volatile uint64_t barrier_count[2];
	uint64_t pad[5];
	RankBlock ranks[MAX_WORLD];
};

struct Comm {
	int rank = 0, world = 1;
	std::string name;
	Shared* sh = nullptr;
	uint8_t* staging[MAX_WORLD] = {};
	size_t staging_size = 0;
	uint64_t barrier_round = 0, seq = 0;
	double timeout = 60.0;
};

thread_local bool g_in_group = false;
struct PendingP2P { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local std::vector<PendingP2P> g_pending;

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char* fmt, ...) {
	char buf[512];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
	fprintf(stderr, "[rccl_loopback pid %d] %s\n", (int)getpid(), buf);
	fflush(stderr);
	return code;
}

int barrier(Comm* c, const char* where) {
	if (c->world == 1) return OK;
	const uint64_t round = c->barrier_round++;
	volatile uint64_t* cnt = &c->sh->barrier_count[round & 1];
	const uint64_t target = (round / 2 + 1) * (uint64_t)c->world;
	__atomic_add_fetch(cnt, 1, __ATOMIC_ACQ_REL);
	const Clock::time_point t0 = Clock::now();
	uint32_t spins = 0;
	while (__atomic_load_n(cnt, __ATOMIC_ACQUIRE) < target) {
		if ((++spins & 0x3ffu) == 0) {
			if (since(t0) > c->timeout) return fail(SYSTEM_ERROR, "rank %d of %d: barrier timed out after %.0f s in %s (collective #%llu) — a rank is missing or issued a different sequence of collectives", c->rank, c->world, c->timeout, where, (unsigned long long)c->seq);
			std::this_thread::yield();
		}
	}
	return OK;
}

#define HIP_OK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail(UNHANDLED, "rank %d: %s failed: %s", c->rank, what, hipGetErrorString(e_)); } while (0)

// post what this rank thinks the collective is; after the barrier: everybody must agree
int post(Comm* c, uint64_t kind, uint64_t count, uint64_t dtype) {
	RankBlock& me = c->sh->ranks[c->rank];
	me.kind = kind; me.count = count; me.dtype = dtype;
	__atomic_store_n(&me.seq, c->seq, __ATOMIC_RELEASE);
	return OK;
}
int agree(Comm* c, const char* what) {
	const RankBlock& me = c->sh->ranks[c->rank];
	for (int q = 0; q < c->world; ++q) {
		const RankBlock& o = c->sh->ranks[q];
		if (o.seq != me.seq || o.kind != me.kind || o.count != me.count || o.dtype != me.dtype)
			return fail(INVALID_USAGE, "rank %d: %s #%llu (count %llu, dtype %llu) but rank %d is in collective #%llu kind %llu (count %llu, dtype %llu)", c->rank, what, (unsigned long long)me.seq,
			            (unsigned long long)me.count, (unsigned long long)me.dtype, q, (unsigned long long)o.seq, (unsigned long long)o.kind, (unsigned long long)o.count, (unsigned long long)o.dtype);
	}
	return OK;
}

struct Ptrs { const void* p[MAX_WORLD]; };

// out[i] = sum over the ranks q = 0 .. world - 1, in that order, of src[q][i]; fp16 is accumulated in fp32 and rounded once
template <typename T, typename ACC>
__global__ void __launch_bounds__(256) sum_ranks_kernel(uint32_t world, uint64_t n, Ptrs src, uint64_t src_offset_elems, T* __restrict__ out) {
	const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
	if (i >= n) return;
	ACC acc = (ACC)((const T*)src.p[0])[src_offset_elems + i];
	for (uint32_t q = 1; q < world; ++q) acc += (ACC)((const T*)src.p[q])[src_offset_elems + i];
	out[i] = (T)acc;
}

int launch_sum(Comm* c, hipStream_t stream, int dtype, uint64_t n, uint64_t src_offset_elems, void* out) {
	Ptrs p;
	for (int q = 0; q < MAX_WORLD; ++q) p.p[q] = q < c->world ? c->staging[q] : nullptr;
	const dim3 grid((uint32_t)((n + 255) / 256)), block(256);
	switch (dtype) {
		case DT_F16: hipLaunchKernelGGL((sum_ranks_kernel<_Float16, float>), grid, block, 0, stream, (uint32_t)c->world, n, p, src_offset_elems, (_Float16*)out); break;
		case DT_F32: hipLaunchKernelGGL((sum_ranks_kernel<float, float>), grid, block, 0, stream, (uint32_t)c->world, n, p, src_offset_elems, (float*)out); break;
		case DT_F64: hipLaunchKernelGGL((sum_ranks_kernel<double, double>), grid, block, 0, stream, (uint32_t)c->world, n, p, src_offset_elems, (double*)out); break;
		case DT_I32: hipLaunchKernelGGL((sum_ranks_kernel<int32_t, int32_t>), grid, block, 0, stream, (uint32_t)c->world, n, p, src_offset_elems, (int32_t*)out); break;
		case DT_U32: hipLaunchKernelGGL((sum_ranks_kernel<uint32_t, uint32_t>), grid, block, 0, stream, (uint32_t)c->world, n, p, src_offset_elems, (uint32_t*)out); break;
		case DT_I64: hipLaunchKernelGGL((sum_ranks_kernel<int64_t, int64_t>), grid, block, 0, stream, (uint32_t)c->world, n, p, src_offset_elems, (int64_t*)out); break;
		case DT_U64: hipLaunchKernelGGL((sum_ranks_kernel<uint64_t, uint64_t>), grid, block, 0, stream, (uint32_t)c->world, n, p, src_offset_elems, (uint64_t*)out); break;
		default: return fail(INVALID_ARGUMENT, "rank %d: reduction of dtype %d is not implemented in the loopback", c->rank, dtype);
	}
	HIP_OK(hipGetLastError(), "sum_ranks_kernel launch");
	return OK;
}

int run_group(std::vector<PendingP2P>& ops) {
	if (ops.empty()) return OK;
	Comm* c = ops[0].comm;
	hipStream_t stream = ops[0].stream;
	for (const PendingP2P& o : ops) if (o.comm != c) return fail(INVALID_USAGE, "a group over several communicators is not implemented in the loopback");
	if (c->world == 1) return fail(INVALID_USAGE, "ncclSend / ncclRecv on a one-rank communicator");
	++c->seq;
	RankBlock& me = c->sh->ranks[c->rank];
	size_t off = 0; uint64_t n_sends = 0;
	for (const PendingP2P& o : ops) {
		if (!o.send) continue;
		if (n_sends >= MAX_P2P) return fail(INVALID_USAGE, "rank %d: more than %d sends in one group", c->rank, MAX_P2P);
		if (off + o.bytes > c->staging_size) return fail(INTERNAL_ERROR, "rank %d: the sends of one group (%zu bytes and more) exceed the staging buffer (NGP_LOOPBACK_STAGING_MB)", c->rank, off + o.bytes);
		HIP_OK(hipMemcpyAsync(c->staging[c->rank] + off, o.buf, o.bytes, hipMemcpyDeviceToDevice, stream), "copy into the staging buffer");
		me.sends[n_sends].peer = (uint32_t)o.peer; me.sends[n_sends].offset = off; me.sends[n_sends].bytes = o.bytes;
		++n_sends; off += (o.bytes + 255) & ~(size_t)255;
	}
	me.n_sends = n_sends;
	post(c, KIND_GROUP, 0, 0);   // (what the peers send is checked per receive below)
	HIP_OK(hipStreamSynchronize(stream), "draining the stream");
	int rc = barrier(c, "ncclGroupEnd (sends staged)"); if (rc) return rc;
	rc = agree(c, "ncclGroupEnd"); if (rc) return rc;
	uint32_t taken[MAX_WORLD] = {};   // the k-th receive from a peer matches that peer's k-th send to this rank
	for (const PendingP2P& o : ops) {
		if (o.send) continue;
		const RankBlock& p = c->sh->ranks[o.peer];
		uint32_t k = 0; const P2P* hit = nullptr;
		for (uint64_t s = 0; s < p.n_sends; ++s) if ((int)p.sends[s].peer == c->rank) { if (k == taken[o.peer]) { hit = &p.sends[s]; break; } ++k; }
		if (!hit) return fail(INVALID_USAGE, "rank %d: receive #%u from rank %d has no matching send", c->rank, taken[o.peer], o.peer);
		if (hit->bytes != o.bytes) return fail(INVALID_USAGE, "rank %d: receive of %zu bytes from rank %d, which sends %llu", c->rank, o.bytes, o.peer, (unsigned long long)hit->bytes);
		++taken[o.peer];
		HIP_OK(hipMemcpyAsync(o.buf, c->staging[o.peer] + hit->offset, o.bytes, hipMemcpyDeviceToDevice, stream), "copy out of a peer's staging buffer");
	}
	HIP_OK(hipStreamSynchronize(stream), "draining the stream");
	return barrier(c, "ncclGroupEnd (receives done)");
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

__attribute__((visibility("default"))) int ncclGetVersion(int* v) { if (v) *v = 0; return OK; }   // 0: "not an RCCL"

__attribute__((visibility("default"))) const char* ncclGetErrorString(int rc) {
	switch (rc) {
		case OK: return "no error (rccl_loopback)";
		case UNHANDLED: return "unhandled HIP error (rccl_loopback: see stderr)";
		case SYSTEM_ERROR: return "system error / time-out (rccl_loopback: see stderr)";
		case INTERNAL_ERROR: return "internal error (rccl_loopback: see stderr)";
		case INVALID_ARGUMENT: return "invalid argument (rccl_loopback: see stderr)";
		case INVALID_USAGE: return "invalid usage: the ranks disagree about a collective (rccl_loopback: see stderr)";
		default: return "unknown result code (rccl_loopback)";
	}
}

__attribute__((visibility("default"))) int ncclGetUniqueId(ncclUniqueId* id) {
	if (!id) return INVALID_ARGUMENT;
	static std::atomic<uint32_t> counter{0};
	std::random_device rd;
	memset(id->internal, 0, sizeof(id->internal));
	snprintf(id->internal, sizeof(id->internal), "/ngp_loopback_%d_%08x_%u", (int)getpid(), (unsigned)rd(), counter.fetch_add(1));
	return OK;
}

__attribute__((visibility("default"))) int ncclCommInitRank(void** out, int world, ncclUniqueId id, int rank) {
	if (!out || world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return fail(INVALID_ARGUMENT, "ncclCommInitRank: rank %d of %d (the loopback carries at most %d ranks)", rank, world, MAX_WORLD);
	id.internal[sizeof(id.internal) - 1] = 0;
	if (strncmp(id.internal, "/ngp_loopback_", 14) != 0) return fail(INVALID_ARGUMENT, "ncclCommInitRank: this unique id was not made by the loopback library (a mix of librccl and the loopback?)");
	Comm* c = new Comm();
	c->rank = rank; c->world = world; c->name = id.internal; c->timeout = timeout_s();
	if (world == 1) { *out = c; return OK; }
	const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);   // whoever comes first creates it; a fresh object is zero-filled
	if (fd < 0) { delete c; return fail(SYSTEM_ERROR, "shm_open(%s) failed", id.internal); }
	if (ftruncate(fd, (off_t)sizeof(Shared)) != 0) { close(fd); delete c; return fail(SYSTEM_ERROR, "ftruncate(%s) failed", id.internal); }
	void* mem = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
	close(fd);
	if (mem == MAP_FAILED) { delete c; return fail(SYSTEM_ERROR, "mmap(%s) failed", id.internal); }
	c->sh = (Shared*)mem;
	c->staging_size = staging_bytes();
	void* mine = nullptr;
	hipError_t e = hipMalloc(&mine, c->staging_size);
	if (e != hipSuccess) { munmap(mem, sizeof(Shared)); delete c; return fail(UNHANDLED, "rank %d: hipMalloc of the %zu-byte staging buffer failed: %s", rank, c->staging_size, hipGetErrorString(e)); }
	c->staging[rank] = (uint8_t*)mine;
	RankBlock& me = c->sh->ranks[rank];
	e = hipIpcGetMemHandle(&me.handle, mine);
	if (e != hipSuccess) { (void)hipFree(mine); munmap(mem, sizeof(Shared)); delete c; return fail(UNHANDLED, "rank %d: hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", rank, hipGetErrorString(e)); }
	__atomic_store_n(&me.ready, MAGIC, __ATOMIC_RELEASE);
	const Clock::time_point t0 = Clock::now();
	for (int q = 0; q < world; ++q) {
		if (q == rank) continue;
		while (__atomic_load_n(&c->sh->ranks[q].ready, __ATOMIC_ACQUIRE) != MAGIC) {
			if (since(t0) > c->timeout) return fail(SYSTEM_ERROR, "rank %d: rank %d never published its staging buffer (%.0f s)", rank, q, c->timeout);
			std::this_thread::sleep_for(std::chrono::microseconds(200));
		}
		void* p = nullptr;
		hipIpcMemHandle_t h;
		memcpy(&h, (const void*)&c->sh->ranks[q].handle, sizeof(h));
		e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
		if (e != hipSuccess) return fail(UNHANDLED, "rank %d: hipIpcOpenMemHandle of rank %d's staging buffer failed: %s", rank, q, hipGetErrorString(e));
		c->staging[q] = (uint8_t*)p;
	}
	const int rc = barrier(c, "ncclCommInitRank");
	if (rc) return rc;
	if (rank == 0) shm_unlink(c->name.c_str());   // everybody has it mapped
	*out = c;
	return OK;
}

__attribute__((visibility("default"))) int ncclCommCount(void* comm, int* n) { if (!comm || !n) return INVALID_ARGUMENT; *n = ((Comm*)comm)->world; return OK; }
__attribute__((visibility("default"))) int ncclCommUserRank(void* comm, int* r) { if (!comm || !r) return INVALID_ARGUMENT; *r = ((Comm*)comm)->rank; return OK; }

// not a collective (the product's shutdown_data_parallel is rank-local by design): behind the last collective's closing barrier nobody reads anybody's staging buffer
__attribute__((visibility("default"))) int ncclCommDestroy(void* comm) {
	Comm* c = (Comm*)comm;
	if (!c) return OK;
	for (int q = 0; q < c->world; ++q) {
		if (!c->staging[q]) continue;
		if (q == c->rank) (void)hipFree(c->staging[q]); else (void)hipIpcCloseMemHandle(c->staging[q]);
	}
	if (c->sh) munmap(c->sh, sizeof(Shared));
	delete c;
	return OK;
}

__attribute__((visibility("default"))) int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
	Comm* c = (Comm*)comm;
	const size_t es = dtype_size(dtype);
	if (!c || !send || !recv || !es) return fail(INVALID_ARGUMENT, "ncclAllReduce: bad argument");
	if (op != 0) return fail(INVALID_ARGUMENT, "ncclAllReduce: only ncclSum is implemented in the loopback");
	if (count == 0) return OK;
	if (c->world == 1) { if (send != recv) HIP_OK(hipMemcpyAsync(recv, send, count * es, hipMemcpyDeviceToDevice, stream), "copy"); return OK; }
	const size_t chunk = c->staging_size / es;
	for (size_t off = 0; off < count; off += chunk) {
		const size_t n = count - off < chunk ? count - off : chunk;
		++c->seq;
		HIP_OK(hipMemcpyAsync(c->staging[c->rank], (const uint8_t*)send + off * es, n * es, hipMemcpyDeviceToDevice, stream), "copy into the staging buffer");
		post(c, KIND_ALLREDUCE, count, (uint64_t)dtype);
		HIP_OK(hipStreamSynchronize(stream), "draining the stream");
		int rc = barrier(c, "ncclAllReduce (staged)"); if (rc) return rc;
		rc = agree(c, "ncclAllReduce"); if (rc) return rc;
		rc = launch_sum(c, stream, dtype, n, 0, (uint8_t*)recv + off * es); if (rc) return rc;
		HIP_OK(hipStreamSynchronize(stream), "draining the stream");
		rc = barrier(c, "ncclAllReduce (summed)"); if (rc) return rc;
	}
	return OK;
}

__attribute__((visibility("default"))) int ncclAllGather(const void* send, void* recv, size_t sendcount, int dtype, void* comm, hipStream_t stream) {
	Comm* c = (Comm*)comm;
	const size_t es = dtype_size(dtype);
	if (!c || !send || !recv || !es) return fail(INVALID_ARGUMENT, "ncclAllGather: bad argument");
	if (sendcount == 0) return OK;
	if (c->world == 1) { if (send != recv) HIP_OK(hipMemcpyAsync(recv, send, sendcount * es, hipMemcpyDeviceToDevice, stream), "copy"); return OK; }
	const size_t chunk = c->staging_size / es;
	for (size_t off = 0; off < sendcount; off += chunk) {
		const size_t n = sendcount - off < chunk ? sendcount - off : chunk;
		++c->seq;
		HIP_OK(hipMemcpyAsync(c->staging[c->rank], (const uint8_t*)send + off * es, n * es, hipMemcpyDeviceToDevice, stream), "copy into the staging buffer");
		post(c, KIND_ALLGATHER, sendcount, (uint64_t)dtype);
		HIP_OK(hipStreamSynchronize(stream), "draining the stream");
		int rc = barrier(c, "ncclAllGather (staged)"); if (rc) return rc;
		rc = agree(c, "ncclAllGather"); if (rc) return rc;
		for (int q = 0; q < c->world; ++q)
			HIP_OK(hipMemcpyAsync((uint8_t*)recv + ((size_t)q * sendcount + off) * es, c->staging[q], n * es, hipMemcpyDeviceToDevice, stream), "copy out of a peer's staging buffer");
		HIP_OK(hipStreamSynchronize(stream), "draining the stream");
		rc = barrier(c, "ncclAllGather (gathered)"); if (rc) return rc;
	}
	return OK;
}

__attribute__((visibility("default"))) int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int dtype, int op, void* comm, hipStream_t stream) {
	Comm* c = (Comm*)comm;
	const size_t es = dtype_size(dtype);
	if (!c || !send || !recv || !es) return fail(INVALID_ARGUMENT, "ncclReduceScatter: bad argument");
	if (op != 0) return fail(INVALID_ARGUMENT, "ncclReduceScatter: only ncclSum is implemented in the loopback");
	if (recvcount == 0) return OK;
	if (c->world == 1) { if (send != recv) HIP_OK(hipMemcpyAsync(recv, send, recvcount * es, hipMemcpyDeviceToDevice, stream), "copy"); return OK; }
	const size_t chunk = c->staging_size / es / (size_t)c->world;   // per chunk: world pieces of `n` elements, piece q = what goes to rank q
	for (size_t off = 0; off < recvcount; off += chunk) {
		const size_t n = recvcount - off < chunk ? recvcount - off : chunk;
		++c->seq;
		for (int q = 0; q < c->world; ++q)
			HIP_OK(hipMemcpyAsync(c->staging[c->rank] + (size_t)q * n * es, (const uint8_t*)send + ((size_t)q * recvcount + off) * es, n * es, hipMemcpyDeviceToDevice, stream), "copy into the staging buffer");
		post(c, KIND_REDUCESCATTER, recvcount, (uint64_t)dtype);
		HIP_OK(hipStreamSynchronize(stream), "draining the stream");
		int rc = barrier(c, "ncclReduceScatter (staged)"); if (rc) return rc;
		rc = agree(c, "ncclReduceScatter"); if (rc) return rc;
		rc = launch_sum(c, stream, dtype, n, (uint64_t)c->rank * n, (uint8_t*)recv + off * es); if (rc) return rc;
		HIP_OK(hipStreamSynchronize(stream), "draining the stream");
		rc = barrier(c, "ncclReduceScatter (summed)"); if (rc) return rc;
	}
	return OK;
}

__attribute__((visibility("default"))) int ncclGroupStart() {
	if (g_in_group) return fail(INVALID_USAGE, "nested ncclGroupStart is not implemented in the loopback");
	g_in_group = true; g_pending.clear();
	return OK;
}
__attribute__((visibility("default"))) int ncclGroupEnd() {
	if (!g_in_group) return fail(INVALID_USAGE, "ncclGroupEnd without ncclGroupStart");
	g_in_group = false;
	std::vector<PendingP2P> ops; ops.swap(g_pending);
	return run_group(ops);
}
static int p2p(bool send, void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) {
	Comm* c = (Comm*)comm;
	const size_t es = dtype_size(dtype);
	if (!c || !buf || !es || peer < 0 || peer >= c->world || peer == c->rank) return fail(INVALID_ARGUMENT, "ncclSend / ncclRecv: bad argument (peer %d)", peer);
	PendingP2P o{send, buf, count * es, peer, c, stream};
	if (g_in_group) { g_pending.push_back(o); return OK; }
	std::vector<PendingP2P> one{o};   // ungrouped: still needs every rank of the communicator to come by (documented above)
	return run_group(one);
}
__attribute__((visibility("default"))) int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) { return p2p(true, (void*)buf, count, dtype, peer, comm, stream); }
__attribute__((visibility("default"))) int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) { return p2p(false, buf, count, dtype, peer, comm, stream); }

}  // extern "C"
