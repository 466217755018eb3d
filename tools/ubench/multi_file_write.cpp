// multi_file_write.cpp — aggregate write rate of F files written AT ONCE, one thread per file, sequential write() of 4 MiB
// pieces from memory (what the pipe's file writers do), on the GPU box's filesystem under the container's CPU quota:
//   g++ -O2 -pthread tools/ubench/multi_file_write.cpp -o /tmp/mfw && /tmp/mfw <dir> [GiB per file]
// (one file takes ~10 GB/s whatever is done; this is the table for 1 .. 16 files the round-3 review asked for.)
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    const size_t per_file = (size_t)((argc > 2 ? atof(argv[2]) : 1.0) * (double)(1ull << 30));
    const size_t PIECE = 4u << 20;
    std::vector<char> src(64u << 20, 'x');
    for (int F : {1, 2, 4, 8, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            std::vector<std::thread> th;
            std::vector<int> ok((size_t)F, 1);
            const double t0 = now();
            for (int f = 0; f < F; ++f)
                th.emplace_back([&, f] {
                    char path[400];
                    snprintf(path, sizeof(path), "%s/mfw_%d.bin", dir, f);
                    const int fd = open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
                    if (fd < 0) { ok[(size_t)f] = 0; return; }
                    size_t done = 0;
                    while (done < per_file) {
                        const size_t k = std::min(PIECE, per_file - done);
                        const ssize_t w = write(fd, src.data() + (done % (src.size() - PIECE)), k);
                        if (w <= 0) { ok[(size_t)f] = 0; break; }
                        done += (size_t)w;
                    }
                    close(fd);
                });
            for (auto& t : th) t.join();
            const double dt = now() - t0;
            bool all = true;
            for (int v : ok) all = all && v;
            printf("%2d files x %.2f GiB, one writer thread each: %.3f s = %6.2f GB/s in total, %5.2f GB/s per file%s\n", F, (double)per_file / (double)(1ull << 30), dt,
                   (double)F * (double)per_file / dt / 1e9, (double)per_file / dt / 1e9, all ? "" : "  (a write FAILED)");
            fflush(stdout);
            for (int f = 0; f < F; ++f) {
                char path[400];
                snprintf(path, sizeof(path), "%s/mfw_%d.bin", dir, f);
                unlink(path);
            }
        }
    }
    return 0;
}
