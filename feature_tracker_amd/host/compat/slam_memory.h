// slam_memory.h — SlamMemory::Malloc / Free (test_optical_flow.cpp:50-51).
#ifndef _SLAM_UTILITY_MEMORY_H_
#define _SLAM_UTILITY_MEMORY_H_
#include <cstdint>
#include <cstdlib>

class SlamMemory {
public:
    static void *Malloc(uint64_t size) { return std::malloc(size); }
    static void Free(void *ptr) { std::free(ptr); }
};
#endif
