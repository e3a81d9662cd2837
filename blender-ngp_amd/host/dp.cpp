// dp.cpp — see dp.h
#include "dp.h"

#include <chrono>
#include <cstring>
#include <stdexcept>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace ngp {

namespace {
constexpr uint64_t MAGIC = 0x6e67705f64703031ull;   // "ngp_dp01"
double now_s() { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); }
struct Slot { double v[3]; volatile uint64_t tag; uint64_t pad[4]; };   // 64 bytes: one cache line per rank and parity
static_assert(sizeof(Slot) == 64, "slot");
}  // namespace

struct ShmCounterExchange::Header {
	volatile uint64_t magic;        // written LAST by rank 0
	double created_at;              // wall clock of the creating rank 0: a segment older than the timeout is a leftover of a crashed run
	uint32_t world, pad0;
	volatile uint64_t blob_ready;
	uint8_t blob[128];
	volatile uint64_t barrier_count[2];
	uint8_t pad1[256 - 8 - 8 - 8 - 8 - 128 - 16];
};

ShmCounterExchange::ShmCounterExchange(uint32_t rank, uint32_t world, const std::string& key, double timeout_s) : m_rank(rank), m_world(world), m_timeout_s(timeout_s) {
	static_assert(sizeof(Header) == 256, "header");
	if (world == 0 || rank >= world) throw std::runtime_error{"ShmCounterExchange: bad rank / world size"};
	m_name = "/ngp_dp_" + key;
	m_bytes = sizeof(Header) + (size_t)2 * world * sizeof(Slot);
	const double t0 = now_s();
	if (rank == 0) {
		shm_unlink(m_name.c_str());   // a leftover of an earlier run, if any
		const int fd = shm_open(m_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
		if (fd < 0) throw std::runtime_error{"ShmCounterExchange: shm_open(" + m_name + ") failed"};
		if (ftruncate(fd, (off_t)m_bytes) != 0) { close(fd); shm_unlink(m_name.c_str()); throw std::runtime_error{"ShmCounterExchange: ftruncate failed"}; }
		m_mem = mmap(nullptr, m_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		if (m_mem == MAP_FAILED) { m_mem = nullptr; shm_unlink(m_name.c_str()); throw std::runtime_error{"ShmCounterExchange: mmap failed"}; }
		memset(m_mem, 0, m_bytes);
		Header* h = (Header*)m_mem;
		h->created_at = t0; h->world = world;
		__atomic_store_n(&h->magic, MAGIC, __ATOMIC_RELEASE);
		return;
	}
	for (;;) {
		const int fd = shm_open(m_name.c_str(), O_RDWR, 0600);
		if (fd >= 0) {
			struct stat st;
			if (fstat(fd, &st) == 0 && (size_t)st.st_size >= m_bytes) {
				void* mem = mmap(nullptr, m_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
				if (mem != MAP_FAILED) {
					Header* h = (Header*)mem;
					// ready, made for this world size, and made by THIS job's rank 0 (not a leftover it has yet to replace)
					if (__atomic_load_n(&h->magic, __ATOMIC_ACQUIRE) == MAGIC && h->world == world && h->created_at >= t0 - m_timeout_s) { m_mem = mem; close(fd); return; }
					munmap(mem, m_bytes);
				}
			}
			close(fd);
		}
		if (now_s() - t0 > m_timeout_s) throw std::runtime_error{"ShmCounterExchange: rank 0's segment " + m_name + " did not appear"};
		std::this_thread::sleep_for(std::chrono::milliseconds(2));
	}
}

ShmCounterExchange::~ShmCounterExchange() {
	if (m_mem) munmap(m_mem, m_bytes);
	if (m_rank == 0) shm_unlink(m_name.c_str());
}

void ShmCounterExchange::all_sum(uint64_t step, const double in[3], double out[3]) {
	Header* h = (Header*)m_mem;
	Slot* slots = (Slot*)(h + 1) + (size_t)(step & 1) * m_world;
	Slot& mine = slots[m_rank];
	mine.v[0] = in[0]; mine.v[1] = in[1]; mine.v[2] = in[2];
	const uint64_t want = step + 1;
	__atomic_store_n(&mine.tag, want, __ATOMIC_RELEASE);   // tag last
	const auto t0 = std::chrono::steady_clock::now();
	uint32_t spins = 0;
	for (uint32_t r = 0; r < m_world; ++r) {
		while (__atomic_load_n(&slots[r].tag, __ATOMIC_ACQUIRE) != want) {
			if ((++spins & 0xfffu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > m_timeout_s)
				throw std::runtime_error{"shared-memory counter exchange timed out at step " + std::to_string(step) + " waiting for rank " + std::to_string(r)};
		}
	}
	out[0] = out[1] = out[2] = 0.0;
	for (uint32_t r = 0; r < m_world; ++r) { out[0] += slots[r].v[0]; out[1] += slots[r].v[1]; out[2] += slots[r].v[2]; }   // same order on every rank: same sums
}

void ShmCounterExchange::publish_blob(const uint8_t blob[128]) {
	Header* h = (Header*)m_mem;
	memcpy(h->blob, blob, 128);
	__atomic_store_n(&h->blob_ready, 1, __ATOMIC_RELEASE);
}
void ShmCounterExchange::fetch_blob(uint8_t blob[128]) {
	Header* h = (Header*)m_mem;
	const double t0 = now_s();
	while (__atomic_load_n(&h->blob_ready, __ATOMIC_ACQUIRE) != 1) {
		if (now_s() - t0 > m_timeout_s) throw std::runtime_error{"ShmCounterExchange: rank 0 never published the communicator id"};
		std::this_thread::sleep_for(std::chrono::milliseconds(1));
	}
	memcpy(blob, h->blob, 128);
}
void ShmCounterExchange::barrier() {
	Header* h = (Header*)m_mem;
	const uint64_t round = m_barrier_round++;
	volatile uint64_t* c = &h->barrier_count[round & 1];
	const uint64_t target = (round / 2 + 1) * m_world;
	__atomic_add_fetch(c, 1, __ATOMIC_ACQ_REL);
	const double t0 = now_s();
	while (__atomic_load_n(c, __ATOMIC_ACQUIRE) < target) {
		if (now_s() - t0 > m_timeout_s) throw std::runtime_error{"ShmCounterExchange: barrier timed out"};
		std::this_thread::sleep_for(std::chrono::microseconds(200));
	}
}

}  // namespace ngp
