// comm_id_cli — drives the id-file rendezvous of device_runtime.h (SharedComm) without a device, for the CPU tests:
//   comm_id_cli publish <path> <byte>          write an id of 128 x <byte> with this process' launch nonce
//   comm_id_cli await   <path> <timeout_ms>    wait for an id carrying this process' launch nonce; prints its first byte
//   comm_id_cli nonce                          prints the launch nonce derived from the environment
//   comm_id_cli optin                          prints whether the environment opts into the sharded path: "off" | "on <rank> <world>" | "error"
// The nonce comes from the environment (FTK_COMM_NONCE, TORCHELASTIC_RUN_ID, MASTER_PORT), as in SharedComm.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "device_runtime.h"

int main(int argc, char **argv) {
    using namespace feature_tracker::device;
    if (argc >= 2 && std::strcmp(argv[1], "nonce") == 0) {
        std::printf("%s\n", CommLaunchNonce().c_str());
        return 0;
    }
    if (argc >= 2 && std::strcmp(argv[1], "optin") == 0) {
        int rank = 0, world = 0;
        std::string error;
        const int state = CommOptIn(&rank, &world, &error);
        if (state < 0) {
            std::printf("error %s\n", error.c_str());
        } else if (state == 0) {
            std::printf("off\n");
        } else {
            std::printf("on %d %d\n", rank, world);
        }
        return 0;
    }
    if (argc == 4 && std::strcmp(argv[1], "publish") == 0) {
        unsigned char id[FTK_UNIQUE_ID_BYTES];
        std::memset(id, std::atoi(argv[3]), sizeof(id));
        std::string error;
        if (!PublishCommId(argv[2], CommLaunchNonce(), id, &error)) {
            std::fprintf(stderr, "%s\n", error.c_str());
            return 1;
        }
        return 0;
    }
    if (argc == 4 && std::strcmp(argv[1], "await") == 0) {
        unsigned char id[FTK_UNIQUE_ID_BYTES];
        std::string error;
        if (!AwaitCommId(argv[2], CommLaunchNonce(), std::atoi(argv[3]), id, &error)) {
            std::fprintf(stderr, "%s\n", error.c_str());
            return 2;
        }
        std::printf("%d\n", static_cast<int>(id[0]));
        return 0;
    }
    std::fprintf(stderr, "usage: comm_id_cli publish <path> <byte> | await <path> <timeout_ms> | nonce | optin\n");
    return 64;
}
