// slam_operations.h — control-flow macros of Slam_Utility/src/operate (optical_flow.cpp:8-9 et al.).
#ifndef _SLAM_UTILITY_OPERATIONS_H_
#define _SLAM_UTILITY_OPERATIONS_H_
#define RETURN_FALSE_IF(condition) \
    if (condition) {               \
        return false;              \
    }
#define RETURN_FALSE_IF_FALSE(condition) \
    if (!(condition)) {                  \
        return false;                    \
    }
#define RETURN_TRUE_IF(condition) \
    if (condition) {              \
        return true;              \
    }
#define RETURN_IF(condition) \
    if (condition) {         \
        return;              \
    }
#define CONTINUE_IF(condition) \
    if (condition) {           \
        continue;              \
    }
#define BREAK_IF(condition) \
    if (condition) {        \
        break;              \
    }
#endif
