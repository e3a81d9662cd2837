// dp.cpp — see dp.h
#include "dp.h"

#include <atomic>
#include <chrono>
#include <random>
#include <cstring>
#include <stdexcept>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace ngp {

namespace {
constexpr uint64_t MAGIC = 0x6e67705f64703032ull;   // "ngp_dp02"
typedef std::chrono::steady_clock Clock;            // monotonic: the time-outs must not depend on wall-clock jumps
double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }
struct Slot { double v[3]; volatile uint64_t tag; uint64_t pad[4]; };   // 64 bytes: one cache line per rank and parity
static_assert(sizeof(Slot) == 64, "slot");
// attach handshake of one rank (below): 64 bytes so that ranks do not share a line
struct Hello { volatile uint64_t nonce, ack, confirmed; uint64_t pad[5]; };
static_assert(sizeof(Hello) == 64, "hello");
uint64_t fresh_nonce() {
	static std::atomic<uint64_t> counter{0};
	std::random_device rd;
	uint64_t n = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)getpid() << 17) ^ (uint64_t)Clock::now().time_since_epoch().count() ^ (counter.fetch_add(1) << 48);
	return n ? n : 1;
}
}  // namespace

struct ShmCounterExchange::Header {
	volatile uint64_t magic;        // written LAST by rank 0
	uint32_t world, pad0;
	volatile uint64_t blob_ready;
	uint8_t blob[128];
	volatile uint64_t barrier_count[2];
	uint8_t pad1[256 - 8 - 8 - 8 - 128 - 16];
};

// Rendezvous.  A name in /dev/shm outlives a crashed job, and the next job with the same key may find the leftover before its own rank 0 has
// replaced it: magic, world size, even a published communicator id all look valid there.  What a leftover cannot do is ANSWER: every attaching
// rank writes a fresh random nonce into its hello slot and keeps the mapping only if a live rank 0 echoes exactly that nonce; otherwise it
// unmaps and opens the name again (by then rank 0 has unlinked the leftover and created this job's segment).  Rank 0 leaves its constructor once
// every rank has confirmed the echo, so a segment is never used by a mix of jobs.  No wall clock is involved.
ShmCounterExchange::ShmCounterExchange(uint32_t rank, uint32_t world, const std::string& key, double timeout_s) : m_rank(rank), m_world(world), m_timeout_s(timeout_s) {
	static_assert(sizeof(Header) == 256, "header");
	if (world == 0 || rank >= world) throw std::runtime_error{"ShmCounterExchange: bad rank / world size"};
	m_name = "/ngp_dp_" + key;
	m_bytes = sizeof(Header) + (size_t)2 * world * sizeof(Slot) + (size_t)world * sizeof(Hello);
	const size_t hello_offset = sizeof(Header) + (size_t)2 * world * sizeof(Slot);
	const Clock::time_point t0 = Clock::now();
	if (rank == 0) {
		shm_unlink(m_name.c_str());   // a leftover of an earlier run, if any
		const int fd = shm_open(m_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
		if (fd < 0) throw std::runtime_error{"ShmCounterExchange: shm_open(" + m_name + ") failed (another live job with the same key?)"};
		if (ftruncate(fd, (off_t)m_bytes) != 0) { close(fd); shm_unlink(m_name.c_str()); throw std::runtime_error{"ShmCounterExchange: ftruncate failed"}; }
		m_mem = mmap(nullptr, m_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		if (m_mem == MAP_FAILED) { m_mem = nullptr; shm_unlink(m_name.c_str()); throw std::runtime_error{"ShmCounterExchange: mmap failed"}; }
		memset(m_mem, 0, m_bytes);
		Header* h = (Header*)m_mem;
		h->world = world;
		__atomic_store_n(&h->magic, MAGIC, __ATOMIC_RELEASE);
		Hello* hello = (Hello*)((uint8_t*)m_mem + hello_offset);
		for (;;) {   // echo every nonce that shows up until each rank has confirmed the one it kept
			uint32_t done = 1;
			for (uint32_t r = 1; r < world; ++r) {
				const uint64_t n = __atomic_load_n(&hello[r].nonce, __ATOMIC_ACQUIRE);
				if (n && __atomic_load_n(&hello[r].ack, __ATOMIC_RELAXED) != n) __atomic_store_n(&hello[r].ack, n, __ATOMIC_RELEASE);
				if (n && __atomic_load_n(&hello[r].confirmed, __ATOMIC_ACQUIRE) == n) ++done;
			}
			if (done == world) return;
			if (since(t0) > m_timeout_s) {
				munmap(m_mem, m_bytes); m_mem = nullptr; shm_unlink(m_name.c_str());
				throw std::runtime_error{"ShmCounterExchange: only " + std::to_string(done) + " of " + std::to_string(world) + " ranks attached to " + m_name};
			}
			std::this_thread::sleep_for(std::chrono::microseconds(200));
		}
	}
	const double patience_s = 0.25;   // how long one mapping is given to answer before the name is opened again
	for (;;) {
		const int fd = shm_open(m_name.c_str(), O_RDWR, 0600);
		if (fd >= 0) {
			struct stat st;
			if (fstat(fd, &st) == 0 && (size_t)st.st_size == m_bytes) {
				void* mem = mmap(nullptr, m_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
				if (mem != MAP_FAILED) {
					Header* h = (Header*)mem;
					if (__atomic_load_n(&h->magic, __ATOMIC_ACQUIRE) == MAGIC && h->world == world) {
						Hello& me = ((Hello*)((uint8_t*)mem + hello_offset))[rank];
						const uint64_t nonce = fresh_nonce();
						__atomic_store_n(&me.nonce, nonce, __ATOMIC_RELEASE);
						const Clock::time_point t1 = Clock::now();
						while (since(t1) < patience_s) {
							if (__atomic_load_n(&me.ack, __ATOMIC_ACQUIRE) == nonce) {
								__atomic_store_n(&me.confirmed, nonce, __ATOMIC_RELEASE);
								m_mem = mem; close(fd);
								return;
							}
							std::this_thread::sleep_for(std::chrono::microseconds(200));
						}
					}
					munmap(mem, m_bytes);
				}
			}
			close(fd);
		}
		if (since(t0) > m_timeout_s) throw std::runtime_error{"ShmCounterExchange: no live rank 0 answered on " + m_name};
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
	const Clock::time_point t0 = Clock::now();
	while (__atomic_load_n(&h->blob_ready, __ATOMIC_ACQUIRE) != 1) {
		if (since(t0) > m_timeout_s) throw std::runtime_error{"ShmCounterExchange: rank 0 never published the communicator id"};
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
	const Clock::time_point t0 = Clock::now();
	while (__atomic_load_n(c, __ATOMIC_ACQUIRE) < target) {
		if (since(t0) > m_timeout_s) throw std::runtime_error{"ShmCounterExchange: barrier timed out"};
		std::this_thread::sleep_for(std::chrono::microseconds(200));
	}
}

}  // namespace ngp
