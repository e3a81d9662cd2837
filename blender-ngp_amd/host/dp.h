// dp.h — host side of the data-parallel training step: one process per GPU of ONE node.  Two exchanges per step:
//   * {samples, compacted samples, loss sum} right after the loss kernel — 24 bytes on the step's critical chain (counters -> rays_per_batch ->
//     next march -> next network pass): POSIX shared memory polled by the hosts, microseconds, no kernel launch, no stream hop;
//   * the fp16 gradient vector between backward and optimizer: RCCL all-reduce over xGMI, ordered on the training stream (ngp_rccl_*).
// The reference has neither (single GPU); the partitioning contract (ray slices, global normalisation) is in DESIGN.md §7.
#pragma once
#include <cstdint>
#include <string>

namespace ngp {

class ShmCounterExchange {
public:
	// A rendezvous of all ranks: rank 0 creates the segment (replacing a stale one of the same name), the others attach and keep the mapping only
	// once a LIVE rank 0 has echoed their random nonce — a leftover of a crashed job never answers, so it is never used (dp.cpp).  Returns on every
	// rank once all have attached; throws std::runtime_error after `timeout_s` (monotonic clock).  `key` should still be unique per job (two LIVE jobs
	// with one key collide: the second rank 0 fails on O_EXCL or replaces the first one's name)
	ShmCounterExchange(uint32_t rank, uint32_t world, const std::string& key, double timeout_s = 60.0);
	~ShmCounterExchange();
	ShmCounterExchange(const ShmCounterExchange&) = delete;
	ShmCounterExchange& operator=(const ShmCounterExchange&) = delete;
	// sum over the ranks of three numbers; `step` must advance by one per call on every rank (two slot parities: a rank is at most one step ahead)
	void all_sum(uint64_t step, const double in[3], double out[3]);
	// 128 opaque bytes from rank 0 to everybody (the RCCL unique id)
	void publish_blob(const uint8_t blob[128]);
	void fetch_blob(uint8_t blob[128]);
	// everybody arrives before anybody leaves (used once, after the communicator exists, so that rank 0 may unlink the name)
	void barrier();
	uint32_t rank() const { return m_rank; }
	uint32_t world() const { return m_world; }
private:
	struct Header;
	uint32_t m_rank, m_world;
	std::string m_name;
	void* m_mem = nullptr;
	size_t m_bytes = 0;
	double m_timeout_s;
	uint64_t m_barrier_round = 0;
};

}  // namespace ngp
