// io_probe.cpp — what the MI355X host gives the pipe's reader / writer threads (run on the GPU box, no GPU needed):
//   g++ -O2 -pthread tools/ubench/io_probe.cpp -o /tmp/io_probe && /tmp/io_probe <dir> [GiB]
// write side: one thread's sequential write(); T threads pwrite()-ing disjoint contiguous ranges of ONE fallocate'd file,
// buffered and O_DIRECT; two files at once.  read side: T threads pread()-ing a page-cached file.  Plus a memcpy ceiling.
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static char* aligned(size_t n) {
    void* p = nullptr;
    if (posix_memalign(&p, 1 << 21, n)) return nullptr;
    memset(p, 'x', n);
    return (char*)p;
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    const size_t N = (size_t)(argc > 2 ? atof(argv[2]) : 2.0) * (1ull << 30);
    const size_t SRC = 256 << 20;
    char* src = aligned(SRC);
    char path[2][300];
    snprintf(path[0], 300, "%s/io_probe_a.bin", dir);
    snprintf(path[1], 300, "%s/io_probe_b.bin", dir);
    auto report = [&](const char* what, int T, double dt, size_t bytes) { printf("%-46s T=%-3d %.3f s  %6.2f GB/s\n", what, T, dt, bytes / dt / 1e9); fflush(stdout); };

    const bool only_mmap = argc > 3;          // io_probe <dir> <GiB> mmap: just the write() baseline and the shared-mapping variants
    // memcpy ceiling
    for (int T : {1, 4, 16, 32}) {
        if (only_mmap) break;
        char* dst = aligned((size_t)T * (64 << 20));
        double t0 = now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { for (int r = 0; r < 8; ++r) memcpy(dst + (size_t)t * (64 << 20), src + (size_t)(t % 4) * (64 << 20), 64 << 20); });
        for (auto& x : th) x.join();
        report("memcpy 64 MiB blocks", T, now() - t0, (size_t)T * 8 * (64 << 20));
        free(dst);
    }
    // 1. one thread, sequential write(), new file
    for (size_t call : {(size_t)1 << 20, (size_t)8 << 20, (size_t)64 << 20}) {
        unlink(path[0]);
        int fd = open(path[0], O_WRONLY | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        for (size_t o = 0; o < N; o += call) (void)!write(fd, src + (o % SRC), call);
        double dt = now() - t0;
        close(fd);
        char w[100]; snprintf(w, 100, "write() sequential, %zu MiB calls", call >> 20);
        report(w, 1, dt, N);
    }
    // 2. two files at once, one thread each
    {
        unlink(path[0]); unlink(path[1]);
        int fd[2] = {open(path[0], O_WRONLY | O_CREAT | O_TRUNC, 0644), open(path[1], O_WRONLY | O_CREAT | O_TRUNC, 0644)};
        double t0 = now();
        std::vector<std::thread> th;
        for (int f = 0; f < 2; ++f) th.emplace_back([&, f] { for (size_t o = 0; o < N; o += 8 << 20) (void)!write(fd[f], src + (o % SRC), 8 << 20); });
        for (auto& x : th) x.join();
        report("two files, write() sequential 8 MiB", 2, now() - t0, 2 * N);
        close(fd[0]); close(fd[1]);
        unlink(path[1]);
    }
    // 3. T threads, disjoint contiguous ranges of one file: buffered (with / without fallocate), O_DIRECT
    for (int mode = 0; mode < 4 && !only_mmap; ++mode) {
        for (int T : {2, 4, 8, 16}) {
            unlink(path[0]);
            int flags = O_WRONLY | O_CREAT | O_TRUNC | (mode >= 2 ? O_DIRECT : 0);
            int fd = open(path[0], flags, 0644);
            if (fd < 0) { printf("mode %d: open failed (%s)\n", mode, strerror(errno)); break; }
            double t0 = now();
            if (mode == 1 || mode == 2) { if (posix_fallocate(fd, 0, N)) printf("fallocate failed\n"); }
            if (mode == 3) { if (ftruncate(fd, N)) printf("ftruncate failed\n"); }
            double tf = now() - t0;
            std::atomic<int> bad{0};
            std::vector<std::thread> th;
            const size_t per = N / T;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (size_t o = 0; o < per; o += 8 << 20) {
                        ssize_t w = pwrite(fd, src + ((t * per + o) % SRC), 8 << 20, t * per + o);
                        if (w != (8 << 20)) { bad++; return; }
                    }
                });
            for (auto& x : th) x.join();
            double dt = now() - t0;
            close(fd);
            const char* nm[4] = {"pwrite ranges, buffered", "pwrite ranges, buffered after fallocate", "pwrite ranges, O_DIRECT after fallocate", "pwrite ranges, O_DIRECT after ftruncate"};
            char w[120]; snprintf(w, 120, "%s%s (prep %.3f)", nm[mode], bad ? " FAILED" : "", tf);
            report(w, T, dt, N);
        }
    }
    // 3b. interleaved 8 MiB pieces (what a chunk-ordered writer would do) O_DIRECT
    for (int T : {4, 8}) {
        if (only_mmap) break;
        unlink(path[0]);
        int fd = open(path[0], O_WRONLY | O_CREAT | O_TRUNC | O_DIRECT, 0644);
        if (fd < 0) break;
        double t0 = now();
        (void)!posix_fallocate(fd, 0, N);
        std::vector<std::thread> th;
        const size_t np = N / (8 << 20);
        for (int t = 0; t < T; ++t) th.emplace_back([&, t] { for (size_t i = t; i < np; i += T) (void)!pwrite(fd, src + ((i * (8 << 20)) % SRC), 8 << 20, i * (8 << 20)); });
        for (auto& x : th) x.join();
        report("pwrite interleaved 8 MiB, O_DIRECT+fallocate", T, now() - t0, N);
        close(fd);
    }
    // 3c. shared mapping of the new file, T threads memcpy disjoint ranges into it (page faults instead of write(): no inode lock)
    for (int mode = 0; mode < 3; ++mode) {
        for (int T : {1, 2, 4, 8, 16}) {
            unlink(path[0]);
            int fd = open(path[0], O_RDWR | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) break;
            double t0 = now();
            if (mode == 1) { if (posix_fallocate(fd, 0, N)) printf("fallocate failed\n"); }
            else if (ftruncate(fd, N)) printf("ftruncate failed\n");
            char* m = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_SHARED | (mode == 2 ? MAP_POPULATE : 0), fd, 0);
            if (m == MAP_FAILED) { printf("mmap failed (%s)\n", strerror(errno)); close(fd); break; }
            double tf = now() - t0;
            std::vector<std::thread> th;
            const size_t per = N / T;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] { for (size_t o = 0; o < per; o += 8 << 20) memcpy(m + t * per + o, src + ((t * per + o) % SRC), 8 << 20); });
            for (auto& x : th) x.join();
            double tc = now() - t0;
            munmap(m, N);
            close(fd);
            double dt = now() - t0;
            const char* nm[3] = {"mmap shared after ftruncate, memcpy ranges", "mmap shared after fallocate, memcpy ranges", "mmap shared + MAP_POPULATE, memcpy ranges"};
            char w[160]; snprintf(w, 160, "%s (prep %.3f, copy done %.3f)", nm[mode], tf, tc);
            report(w, T, dt, N);
        }
    }
    if (argc > 3) { unlink(path[0]); return 0; }
    // 4. read side: the file is in the page cache now?  write it buffered first
    {
        unlink(path[0]);
        int fd = open(path[0], O_WRONLY | O_CREAT | O_TRUNC, 0644);
        for (size_t o = 0; o < N; o += 8 << 20) (void)!write(fd, src + (o % SRC), 8 << 20);
        close(fd);
        char* dst = aligned(N);
        for (int T : {1, 2, 4, 8, 16, 32, 64}) {
            fd = open(path[0], O_RDONLY);
            double t0 = now();
            std::vector<std::thread> th;
            const size_t np = N / (4 << 20);
            std::atomic<size_t> next{0};
            for (int t = 0; t < T; ++t) th.emplace_back([&] { for (;;) { size_t i = next.fetch_add(1); if (i >= np) break; (void)!pread(fd, dst + i * (4 << 20), 4 << 20, i * (4 << 20)); } });
            for (auto& x : th) x.join();
            report("pread 4 MiB pieces from the page cache", T, now() - t0, N);
            close(fd);
        }
        free(dst);
    }
    unlink(path[0]);
    return 0;
}
