// The exchange a sharded proof makes per round, measured in C++ (no Python between the calls): `world` processes (fork) exchange 64-byte
// records through the shared-memory board of csrc/shard_group.hpp.   g++ -O2 -std=c++17 -I. tools/exp_board.cpp -o /tmp/exp_board -lrt
// prints  world <w>: <us per all-gather>   for w = 2, 4, 8   (tools/exp_collective.py runs it and adds the gloo / RCCL figures)
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "jolt-atlas_amd/csrc/shard_group.hpp"

int main() {
    const int N = 200000;
    for (int world : {2, 4, 8}) {
        char name[64];
        snprintf(name, sizeof name, "/atlas_expb_%d_%d", (int)getpid(), world);
        int rank = 0;
        for (int r = 1; r < world; r++) { if (fork() == 0) { rank = r; break; } }
        atlas_shard_group g;
        if (!g.open(name, world, rank)) { fprintf(stderr, "open failed\n"); _exit(1); }
        uint64_t rec[8] = {(uint64_t)rank, 1, 2, 3, 4, 5, 6, 7}, all[8 * 64];
        for (int i = 0; i < 2000; i++) g.allgather(rec, 64, all);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) { rec[1] = all[8 * ((rank + 1) % world) + 1] + 1; g.allgather(rec, 64, all); }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        g.close();
        if (rank) _exit(0);
        while (wait(nullptr) > 0) {}
        printf("world %d: %.3f us per all-gather of 64-byte records\n", world, us);
    }
    return 0;
}
