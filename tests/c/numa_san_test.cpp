// numa.hpp under ASan / UBSan: the cpulist parser on what sysfs prints (and on garbage), thread affinity restricted to a subset
// of what the process is allowed and restored, the mempolicy scope, move_pages in query mode -- every call must either work or
// say "nothing done", on a box with one memory node as on one with two.
#include "../../garage_amd/csrc/numa.hpp"

#include <cstdlib>
#include <thread>

#define CHECK(x)                                                                      \
	do {                                                                          \
		if (!(x)) {                                                           \
			std::fprintf(stderr, "%s:%d: CHECK(%s)\n", __FILE__, __LINE__, #x); \
			std::exit(1);                                                 \
		}                                                                     \
	} while (0)

int main()
{
	using namespace gecnuma;
	CHECK((parse_cpulist("0-3\n") == std::vector<int>{0, 1, 2, 3}));
	CHECK((parse_cpulist("0-1,128-129") == std::vector<int>{0, 1, 128, 129}));
	CHECK((parse_cpulist("7") == std::vector<int>{7}));
	CHECK(parse_cpulist("").empty() && parse_cpulist("\n").empty() && parse_cpulist("abc").empty());
	CHECK((parse_cpulist("5-3") == std::vector<int>{}));                       // a reversed range is empty, not a wrap
	CHECK(parse_cpulist("0-99999999999").size() == 4096);                        // clamped
	CHECK(node_of_pci(nullptr) == -1 && node_of_pci("") == -1 && node_of_pci("ffff:ff:1f.7") == -1);
	CHECK(cpus_of_node(-1).empty() && cpus_of_node(100000).empty());
	const int nn = node_count();
	CHECK(nn >= 0);
	if (nn > 0)
		CHECK(!cpus_of_node(0).empty());
	// affinity: a thread restricted to ONE of the allowed CPUs sees exactly that; an empty / disjoint set changes nothing
	const std::vector<int> allowed = affinity_of_this_thread();
	CHECK(!allowed.empty());
	std::thread([&] {
		CHECK(!bind_this_thread({}));
		CHECK(!bind_this_thread({100000}));
		CHECK(affinity_of_this_thread() == allowed);
		CHECK(bind_this_thread({allowed.back(), 100000}));
		CHECK((affinity_of_this_thread() == std::vector<int>{allowed.back()}));
	}).join();
	CHECK(affinity_of_this_thread() == allowed);                                // the calling thread was never touched
	// memory: a page touched under ScopedBind(0) is on node 0 when the kernel allows the calls; -1 otherwise
	{
		ScopedBind none(-1);
		CHECK(!none.ok());
		ScopedBind b(0);
		std::vector<char> v(1 << 16, 1);
		const int at = node_of_address(v.data() + 8192);
		CHECK(at == -1 || (at >= 0 && at < (nn > 0 ? nn : 1)));
		if (b.ok() && at >= 0)
			CHECK(at == 0);
	}
	int local = 0;
	(void)node_of_address(&local);
	std::puts("numa_san_test ok");
	return 0;
}
