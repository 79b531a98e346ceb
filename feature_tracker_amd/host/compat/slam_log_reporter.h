// slam_log_reporter.h — stand-in for Slam_Utility's log macros (stream syntax, colours).
#ifndef _SLAM_UTILITY_LOG_REPORTER_H_
#define _SLAM_UTILITY_LOG_REPORTER_H_
#include <iostream>
#include <sstream>
#include <string>

#include "basic_type.h"

#define RESET_COLOR "\033[0m"
#define BLACK "\033[30m"
#define RED "\033[31m"
#define GREEN "\033[32m"
#define YELLOW "\033[33m"
#define BLUE "\033[34m"
#define MAGENTA "\033[35m"
#define CYAN "\033[36m"
#define WHITE "\033[37m"

#define ReportText(...) std::cout << __VA_ARGS__
#define ReportInfo(...) std::cout << GREEN "[Info] " RESET_COLOR << __VA_ARGS__ << std::endl
#define ReportDebug(...) std::cout << CYAN "[Debug] " RESET_COLOR << __VA_ARGS__ << std::endl
#define ReportWarn(...) std::cout << YELLOW "[Warn] " RESET_COLOR << __VA_ARGS__ << std::endl
#define ReportError(...) std::cerr << RED "[Error] " RESET_COLOR << __VA_ARGS__ << std::endl
// LogVec / LogQuat (test_direct_method.cpp:96): printable forms of small vectors and quaternions
template <int R, int C>
inline std::string LogVec(const FixedMat<R, C> &v) {
    std::ostringstream os;
    os << "[";
    for (int i = 0; i < R * C; ++i) {
        os << (i ? ", " : "") << v(i);
    }
    os << "]";
    return os.str();
}
inline std::string LogQuat(const Quat &q) {
    std::ostringstream os;
    os << "[wxyz][" << q.w() << ", " << q.x() << ", " << q.y() << ", " << q.z() << "]";
    return os.str();
}
#endif
