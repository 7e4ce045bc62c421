// numa.hpp -- where a device lane's host side runs and keeps its memory.  Header-only, Linux only, no libnuma
// (the image has none): sysfs for the topology, pthread affinity for threads, set_mempolicy / move_pages through syscall(2) for
// memory.  Everything degrades to "do nothing": a box with one node, a container that forbids the calls, a device whose sysfs entry
// says -1 -- the caller then runs exactly as before.
//
// Why it matters here: the deployable paths are host-fed.  A lane's pinned staging slots and shard buffers are what the device's
// DMA engines and link kernels read, and what the lane's copy / pool / batcher threads fill and check; on a two-socket node half the
// GPUs hang off each socket, and a thread or a page on the other socket turns every one of those accesses into a trip over the
// inter-socket fabric.  (No reference counterpart: Garage has no device.  The closest it has is sharding its own resources per
// use -- the data-dir shards of src/block/manager.rs:679-689 and the RAM buffer permits of :156, 380-385.)
#pragma once

#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace gecnuma {

// "0-63,128-191" -> {0..63, 128..191}
inline std::vector<int> parse_cpulist(const std::string &s)
{
	std::vector<int> cpus;
	size_t i = 0;
	while (i < s.size()) {
		while (i < s.size() && !std::isdigit((unsigned char)s[i]))
			++i;
		if (i >= s.size())
			break;
		long a = 0;
		while (i < s.size() && std::isdigit((unsigned char)s[i]))
			a = a * 10 + (s[i++] - '0');
		long b = a;
		if (i < s.size() && s[i] == '-') {
			++i;
			b = 0;
			while (i < s.size() && std::isdigit((unsigned char)s[i]))
				b = b * 10 + (s[i++] - '0');
		}
		for (long c = a; c <= b && c < 4096; ++c)
			cpus.push_back((int)c);
	}
	return cpus;
}

inline bool read_small_file(const std::string &path, std::string &out)
{
	FILE *f = std::fopen(path.c_str(), "r");
	if (!f)
		return false;
	char buf[4096];
	const size_t n = std::fread(buf, 1, sizeof buf - 1, f);
	std::fclose(f);
	out.assign(buf, n);
	return true;
}

// how many memory nodes the kernel shows (1 on a box without NUMA, 0 if sysfs is not there)
inline int node_count()
{
	int n = 0;
	std::string s;
	while (n < 1024 && read_small_file("/sys/devices/system/node/node" + std::to_string(n) + "/cpulist", s))
		++n;
	return n;
}

inline std::vector<int> cpus_of_node(int node)
{
	std::string s;
	if (node < 0 || !read_small_file("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist", s))
		return {};
	return parse_cpulist(s);
}

// the node a PCI device hangs off ("0000:23:00.0", any case); -1 = unknown (no sysfs entry, or the platform says -1)
inline int node_of_pci(const char *bdf)
{
	std::string id(bdf ? bdf : "");
	for (char &c : id)
		c = (char)std::tolower((unsigned char)c);
	std::string s;
	if (id.empty() || !read_small_file("/sys/bus/pci/devices/" + id + "/numa_node", s))
		return -1;
	return std::atoi(s.c_str());
}

// Restrict the calling thread to `cpus` (intersected with what the process is allowed: a cgroup / taskset narrower than the node
// is respected).  false = nothing changed (empty set, empty intersection, or the call failed).
inline bool bind_this_thread(const std::vector<int> &cpus)
{
	if (cpus.empty())
		return false;
	cpu_set_t allowed, want;
	CPU_ZERO(&allowed);
	if (sched_getaffinity(0, sizeof allowed, &allowed) != 0)
		return false;
	CPU_ZERO(&want);
	int n = 0;
	for (int c : cpus)
		if (c >= 0 && c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) {
			CPU_SET(c, &want);
			++n;
		}
	if (n == 0)
		return false;
	return pthread_setaffinity_np(pthread_self(), sizeof want, &want) == 0;
}

// the CPUs the calling thread may run on
inline std::vector<int> affinity_of_this_thread()
{
	std::vector<int> out;
	cpu_set_t s;
	CPU_ZERO(&s);
	if (sched_getaffinity(0, sizeof s, &s) != 0)
		return out;
	for (int c = 0; c < CPU_SETSIZE; ++c)
		if (CPU_ISSET(c, &s))
			out.push_back(c);
	return out;
}

// the node the page at `p` is on (move_pages in query mode); -1 = not resident / not allowed / not Linux
inline int node_of_address(const void *p)
{
#ifdef SYS_move_pages
	void *page = reinterpret_cast<void *>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)4095);
	int status = -1;
	if (syscall(SYS_move_pages, 0, 1ul, &page, nullptr, &status, 0) != 0)
		return -1;
	return status;
#else
	(void)p;
	return -1;
#endif
}

// While alive, pages the calling thread faults in (or a driver pins on its behalf) come from `node` (MPOL_PREFERRED: from that node
// while it has memory, from the others when it is full -- a lane must not fail to allocate because ITS socket is the busy one); the
// default policy is restored on the way out.  node < 0 = no-op.  ok() says whether the kernel took it.
class ScopedBind {
public:
	explicit ScopedBind(int node)
	{
#ifdef SYS_set_mempolicy
		if (node < 0 || node >= 1024)
			return;
		unsigned long mask[16] = {0};
		mask[node / (8 * sizeof(unsigned long))] = 1ul << (node % (8 * sizeof(unsigned long)));
		set_ = syscall(SYS_set_mempolicy, /*MPOL_PREFERRED*/ 1, mask, sizeof mask * 8) == 0;
#else
		(void)node;
#endif
	}
	~ScopedBind()
	{
#ifdef SYS_set_mempolicy
		if (set_)
			(void)syscall(SYS_set_mempolicy, /*MPOL_DEFAULT*/ 0, nullptr, 0);
#endif
	}
	bool ok() const { return set_; }
	ScopedBind(const ScopedBind &) = delete;
	ScopedBind &operator=(const ScopedBind &) = delete;

private:
	bool set_ = false;
};

}  // namespace gecnuma
