// tick_tock.h — wall-clock timer (test_optical_flow.cpp:69-73).
#ifndef _SLAM_UTILITY_TICK_TOCK_H_
#define _SLAM_UTILITY_TICK_TOCK_H_
#include <chrono>

class TickTock {
public:
    TickTock() { TickInMillisecond(); }
    void TickInMillisecond() { start_ = std::chrono::steady_clock::now(); }
    float TockInMillisecond() const { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - start_).count(); }
    float TockTickInMillisecond() {
        const float t = TockInMillisecond();
        TickInMillisecond();
        return t;
    }
    float TockInSecond() const { return TockInMillisecond() * 1e-3f; }

private:
    std::chrono::steady_clock::time_point start_;
};
#endif
