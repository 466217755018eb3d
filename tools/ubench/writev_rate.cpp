// How fast does ONE thread get a big file into the page cache — write() of large pieces against writev() of the pieces the pipe's
// spans mode writes (13 KB runs of whole records with a record-sized gap between them, 1024 iovecs per call) — and does it matter
// which NUMA node the source buffer was first touched on?  (round 5: file -> file ran at 7.6 GB/s per good file with writev from the
// input ring against 10.4 GB/s with write() from the output sets)
//   writev_rate DIR MiB
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/uio.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool bind_node(int node) {
    char path[96], list[4096] = "";
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    const bool got = fgets(list, sizeof(list), f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t want; CPU_ZERO(&want);
    for (char* p = list; *p;) {
        char* q; long a = strtol(p, &q, 10); if (q == p) break; long b = a;
        if (*q == '-') { p = q + 1; b = strtol(p, &q, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, &want);
        p = *q == ',' ? q + 1 : q; if (*q != ',') break;
    }
    return sched_setaffinity(0, sizeof(want), &want) == 0;
}
static uint8_t* alloc_on(int node, size_t n) {
    uint8_t* p = nullptr;
    std::thread t([&] {
        if (node >= 0) bind_node(node);
        p = (uint8_t*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        madvise(p, n, MADV_HUGEPAGE);
        memset(p, 'A', n);
    });
    t.join();
    return p;
}
static double run(const char* dir, const uint8_t* src, size_t n, int mode, int wnode, size_t seg, size_t gap) {
    double dt = 0;
    std::thread t([&] {
        if (wnode >= 0) bind_node(wnode);
        char path[256]; snprintf(path, 256, "%s/wv.bin", dir);
        int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        const double t0 = now();
        if (mode == 0) {
            for (size_t o = 0; o < n;) { ssize_t w = write(fd, src + o, std::min<size_t>(n - o, 1u << 30)); if (w <= 0) break; o += (size_t)w; }
        } else {
            std::vector<iovec> iov;
            for (size_t o = 0; o + seg <= n; o += seg + gap) iov.push_back({(void*)(src + o), seg});
            for (size_t i = 0; i < iov.size();) {
                const int cnt = (int)std::min<size_t>(iov.size() - i, 1024);
                ssize_t w = writev(fd, iov.data() + i, cnt);
                if (w <= 0) break;
                while (w > 0 && i < iov.size()) {
                    if ((size_t)w >= iov[i].iov_len) { w -= (ssize_t)iov[i].iov_len; ++i; }
                    else { iov[i].iov_base = (uint8_t*)iov[i].iov_base + w; iov[i].iov_len -= (size_t)w; w = 0; }
                }
            }
        }
        dt = now() - t0;
        close(fd); unlink(path);
    });
    t.join();
    return dt;
}
int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    const size_t n = (size_t)(argc > 2 ? atoll(argv[2]) : 1700) << 20;
    int nodes = 0;
    for (; nodes < 8; ++nodes) { char p[96]; snprintf(p, 96, "/sys/devices/system/node/node%d/cpulist", nodes); if (access(p, R_OK)) break; }
    printf("%d NUMA node(s); %zu MiB per run, one writer thread\n", nodes, n >> 20);
    for (int an = 0; an < (nodes > 1 ? 2 : 1); ++an) {
        uint8_t* src = alloc_on(nodes > 1 ? an : -1, n);
        for (int wn = 0; wn < (nodes > 1 ? 2 : 1); ++wn) {
            for (int rep = 0; rep < 2; ++rep) {
                const double a = run(dir, src, n, 0, nodes > 1 ? wn : -1, 0, 0);
                const double b = run(dir, src, n, 1, nodes > 1 ? wn : -1, 13000, 347);
                const double c = run(dir, src, n, 1, nodes > 1 ? wn : -1, 1 << 20, 347);
                const double d = run(dir, src, n, 1, nodes > 1 ? wn : -1, 3470, 347);
                printf("buffer on node %d, writer on node %d: write() %.2f GB/s | writev 13 KB pieces %.2f | writev 1 MiB pieces %.2f | writev 3.4 KB pieces %.2f GB/s\n", an, wn,
                       n / a / 1e9, n * (13000.0 / 13347) / b / 1e9, n / c / 1e9, n * (3470.0 / 3817) / d / 1e9);
            }
        }
        munmap(src, n);
    }
    return 0;
}
